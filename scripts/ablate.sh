#!/bin/bash
# Diagnostic: time the segmentation with one ingredient of the conv kernels removed (results are wrong, timing only).
# Runs on the GPU box: builds a private copy of the library per ablation into /tmp and points the loader at it.
set -e
cd "$(dirname "$0")/.."
for ab in 0 1 2 4 7; do
  rm -rf /tmp/abl_$ab && mkdir -p /tmp/abl_$ab
  for f in oai_analysis_2_amd/csrc/*.hip oai_analysis_2_amd/csrc/*.cpp; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -DOAI_ABLATE=$ab -x hip -c $f -o /tmp/abl_$ab/$(basename $f).o &
  done
  wait
  hipcc -shared -fPIC --offload-arch=gfx950 /tmp/abl_$ab/*.o -o /tmp/abl_$ab/liboai_hip.so
  for prec in f32 bf16x6; do
    OAI_LIB_PATH=/tmp/abl_$ab/liboai_hip.so PREC=$prec python scripts/perf_layers.py | tail -1 | sed "s/^/ablate=$ab /"
  done
done
