#!/bin/bash
# per-layer kernel trace of the exact-fp32 pass, Winograd form and direct form, same box: bash scripts/prof_f32_layers.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/f32_layers; rm -rf $O; mkdir -p $O; cd $R
export PREC=f32 REPS=2
for w in 1 0; do
  export OPTIONS=winograd_f32=$w
  timeout 400 rocprofv3 --kernel-trace -d $O/w$w -o t --output-format csv -- python3 scripts/trace_layers.py > $O/w$w.log 2>&1
  echo "== winograd_f32=$w"; tail -1 $O/w$w.log
  python3 scripts/per_layer_table_f32.py $(find $O/w$w -name "*kernel_trace.csv" | head -1) | tee $O/table_w$w.md
done
