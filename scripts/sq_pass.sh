#!/bin/bash
# SQ / GRBM counters of the segmentation kernels (own pass: counters + kernel-trace only) for each OPTIONS string: bash scripts/sq_pass.sh "wide=0" "wide=1"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq_pass; rm -rf $O; mkdir -p $O; cd $R
export PREC=${PREC:-fp16x3}
i=0
for opt in "$@"; do
  export OPTIONS="$opt"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/s$i -o s --output-format csv -- python3 scripts/perf_layers.py > $O/s$i.log 2>&1
  echo "== OPTIONS=$opt"; python3 scripts/sq_summary.py $O/s$i 4 | tee $O/sq_$i.md
  i=$((i+1))
done
