#!/bin/bash
# round 5: the LDS-staged brick form of grid_sample3d / compose against the gather form: bit-identity test, time (resident and 8 sets cycled), TCP counters
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_warp; rm -rf $O; mkdir -p $O; cd $R
python -m pytest tests/test_warp_gpu.py -q 2>&1 | tail -4 > $O/test.log
for b in 0 1; do BRICK=$b python3 scripts/bench_warp.py 2>&1 | grep -v "^{" > $O/bench_brick$b.log; done
for b in 0 1; do
  i=0
  for set in "TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES TCP_GATE_EN1 TCP_GATE_EN2" "TCP_TCC_READ_REQ TCP_TOTAL_ACCESSES TCP_TA_DATA_STALL_CYCLES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    BRICK=$b ROT=2 rocprofv3 --kernel-trace --pmc $set -d $O/b${b}p$i -o p --output-format csv -- python3 scripts/bench_warp.py > $O/b${b}p$i.log 2>&1 || echo "set $i failed"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_warp"
with open(O + "/summary.txt", "w") as fo:
    for b in (0, 1):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
        for f in glob.glob(O + f"/b{b}p*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"][:72]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
        fo.write(f"== brick {b}\n")
        for k, v in agg.items():
            if "sample_" not in k: continue
            fo.write(k + "\n")
            for c, x in sorted(v.items()):
                fo.write(f"   {c:36s} {x / cnt[k][c]:.4g} per launch ({cnt[k][c]} launches)\n")
print(open(O + "/summary.txt").read()[:6000])
PY
cat $O/test.log $O/bench_brick0.log $O/bench_brick1.log
