#!/bin/bash
# diagnostic library: where conv3_wino_sres spends its time -- timing-only switches (OAI_DBG bits, results wrong), per-layer table of one pass each
ulimit -c 0
export OAI_LIB_PATH=$GRAFT_REPO_ROOT/build/diag/liboai_hip_diag.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wino_var; rm -rf $O; mkdir -p $O; cd $R
for v in "$@"; do
  export OAI_DBG=$v
  rocprofv3 --kernel-trace -d $O/t$v -o t --output-format csv -- python3 scripts/trace_layers.py > $O/t$v.log 2>&1
  f=$(find $O/t$v -name "*kernel_trace.csv" | head -1)
  echo "== OAI_DBG=$v $(tail -1 $O/t$v.log)"
  python3 scripts/per_layer_table.py $f | grep -E "ec2|ec4|dc8|dc5|dc2|sum"
done
