#!/bin/bash
# per-layer table of one full-size pass for each OPTIONS string given:  bash scripts/layers_ab.sh "wide=0" "wide=1"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/layers_ab; rm -rf $O; mkdir -p $O; cd $R
i=0
for opt in "$@"; do
  export OPTIONS="$opt"
  rocprofv3 --kernel-trace -d $O/t$i -o t --output-format csv -- python3 scripts/trace_layers.py > $O/t$i.log 2>&1
  f=$(find $O/t$i -name "*kernel_trace.csv" | head -1)
  echo "== OPTIONS=$opt"; tail -1 $O/t$i.log
  python3 scripts/per_layer_table.py $f > $O/table_$i.md; cat $O/table_$i.md
  i=$((i+1))
done
