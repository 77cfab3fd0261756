#!/bin/bash
# diagnostic library: timing of the conv3_igemm_sres2 variants (OAI_WIDE_VAR) against wide=0, full-size segmentation
export OAI_LIB_PATH=$GRAFT_REPO_ROOT/build/diag/liboai_hip_diag.so
for v in "$@"; do
  echo "== OAI_WIDE_VAR=$v"; OAI_WIDE_VAR=$v python scripts/ab_option.py wide 0,1 2 2>&1 | tail -4
done
