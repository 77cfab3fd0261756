#!/bin/bash
# per-layer table of one full-size pass for the shipped library and each experimental library (scripts/build_exp.py): bash scripts/exp_layers.sh 1 3 7
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/exp_layers; rm -rf $O; mkdir -p $O; cd $R
for b in 0 "$@"; do
  if [ "$b" != "0" ]; then export OAI_LIB_PATH=$R/build/exp/liboai_hip_exp$b.so; else unset OAI_LIB_PATH; fi
  rocprofv3 --kernel-trace -d $O/t$b -o t --output-format csv -- python3 scripts/trace_layers.py > $O/t$b.log 2>&1
  f=$(find $O/t$b -name "*kernel_trace.csv" | head -1)
  echo "== OAI_EXP=$b"; python3 scripts/per_layer_table.py $f | grep -E "ec3|ec4|ec5|dc8|dc7|dc5|dc4|dc2|sum"
done
