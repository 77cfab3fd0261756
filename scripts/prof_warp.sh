#!/bin/bash
# rocprofv3 kernel-trace of the warp bench: per-kernel average durations (the authoritative kernel-only numbers)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/warp; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O -o w --output-format csv -- python3 scripts/bench_warp.py > $O/bench.log 2>&1
head -5 $O/bench.log
python3 - <<'PY'
import csv, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/warp"
rows = list(csv.DictReader(open(O + "/w_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# group consecutive runs of the same kernel (each bench case = 55 launches)
runs = []
for r in rows:
    k = r["Kernel_Name"][:60]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if runs and runs[-1][0] == k and len(runs[-1][1]) < 55: runs[-1][1].append(d)
    else: runs.append((k, [d]))
for k, ds in runs:
    if len(ds) >= 50: print(f"{k:60s} n={len(ds)} avg of last 50: {sum(ds[-50:])/50:7.2f} us  min {min(ds):7.2f}")
PY
