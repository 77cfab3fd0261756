#!/bin/bash
# (a TA_* counter set -- TA_TA_BUSY TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES -- aborts rocprofv3 on this image and
# hangs until the time limit: do not add it back)
# PMC: L1 (TCP) counters of the 160^3 image warp and compose kernels (scripts/bench_warp.py).  Counters + kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_warp; mkdir -p $O; cd $R
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES TCP_GATE_EN1 TCP_GATE_EN2" \
           "TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TOTAL_ACCESSES TCP_TA_DATA_STALL_CYCLES" \
           "TCP_TCP_TA_DATA_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p$i --output-format csv -- python3 scripts/bench_warp.py > $O/p$i.log 2>&1 || echo "set $i failed: $set"
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_warp"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:64]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open(O + "/summary.txt", "w") as fo:
    for k, v in agg.items():
        if "sample_kernel" not in k: continue
        fo.write(k + "\n")
        for c, x in sorted(v.items()):
            fo.write(f"   {c:36s} {x / cnt[k][c]:.4g} per launch ({cnt[k][c]} launches)\n")
print(open(O + "/summary.txt").read()[:5000])
PY
