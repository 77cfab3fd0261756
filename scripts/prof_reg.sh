#!/bin/bash
# registration-side profile: warp bench (K14/K15 at 160^3), config-3 loop op-by-op vs fused, resample, ICON direction; kernel trace + stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/reg; mkdir -p $O; cd $R
python3 scripts/bench_warp.py > $O/warp.log 2>&1; tail -6 $O/warp.log
rocprofv3 --kernel-trace --stats -d $O/loop -o loop --output-format csv -- python3 scripts/bench_reg_loop.py > $O/loop.log 2>&1
grep -v "^{" $O/loop.log | tail -12
f=$(find $O/loop -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-170
