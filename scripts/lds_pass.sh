#!/bin/bash
# LDS counters of the segmentation kernels (counters + kernel trace only): bank conflicts of the Winograd transform / A-fragment reads
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lds; rm -rf $O; mkdir -p $O; cd $R
export PREC=fp16x3 TILES=160
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d $O/p -o l --output-format csv -- python3 scripts/perf_layers.py > $O/l.log 2>&1
tail -2 $O/l.log
python3 - "$O" <<'PY'
import csv, glob, os, sys, collections
O = sys.argv[1]
f = glob.glob(os.path.join(O, "p", "**", "*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"][:58]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:9]:
    act = v.get("SQ_LDS_IDX_ACTIVE", 0)
    print(f"{k:58s} conflict/active {v.get('SQ_LDS_BANK_CONFLICT', 0) / act if act else 0:.3f}  LDS active / wave cycles {v.get('SQ_ACTIVE_INST_LDS', 0) / max(v.get('SQ_WAVE_CYCLES', 1), 1):.3f}  insts {v.get('SQ_INSTS_LDS', 0):.3g}")
PY
