#!/bin/bash
# PMC passes over the per-layer perf script (fp16x3 split-resident path). Counters only, kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_sres
mkdir -p $O
cd $R; export PREC=fp16x3
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O/p$i -o p$i --output-format csv -- python3 scripts/perf_layers.py > $O/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_sres"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))
with open(O + "/summary.txt", "w") as fo:
    for k, v in rows[:12]:
        fo.write(k + "\n")
        for c, x in sorted(v.items()):
            fo.write(f"   {c:32s} {x:.4g}\n")
print(open(O + "/summary.txt").read()[:6000])
PY
