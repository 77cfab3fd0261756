#!/bin/bash
# memory-side counters of the conv3_wino_f32 ablation probe (own passes: counters + kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wino_f32_pmc; rm -rf $O; mkdir -p $O
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -Xclang -target-feature -Xclang -packed-fp32-ops -I$R/include $R/scripts/micro/wino_f32_ablate.hip -o /tmp/wino_f32_ablate 2> $O/build.err || { tail -5 $O/build.err; exit 1; }
i=0
for set in "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY"; do
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o s --output-format csv -- /tmp/wino_f32_ablate > $O/p$i.log 2>&1
  python3 - $O/p$i <<'PY'
import collections, csv, glob, os, sys
O = sys.argv[1]
f = glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True)
if not f: print("no counters in", O); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"][:64]][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for v in agg.values() for c in v})
print("kernel".ljust(64), *[c[:26].rjust(27) for c in names])
for k, v in agg.items(): print(k.ljust(64), *[f"{v.get(c, 0):27.4g}" for c in names])
PY
  i=$((i+1))
done
