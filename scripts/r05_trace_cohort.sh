#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_cohort_trace; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --memory-copy-trace -d $O -o c --output-format csv -- python3 scripts/trace_cohort.py > $O/run.log 2>&1
ls $O
python3 - <<'PY'
import csv, os, glob
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05_cohort_trace"
k = sorted(csv.DictReader(open(glob.glob(O + "/*kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
m = sorted(csv.DictReader(open(glob.glob(O + "/*memory_copy_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
print("copy columns:", list(m[0].keys()))
# volume boundaries: the ec0-fused ec1 kernel starts a volume's segmentation
firsts = [i for i, r in enumerate(k) if "conv3_igemm_sres<4, 16, 2, 4, 1, false, true" in r["Kernel_Name"]]
t0 = int(k[firsts[-5]]["Start_Timestamp"])
ev = []
for i in firsts[-5:]:
    prev = k[i - 1]
    ev.append((int(prev["End_Timestamp"]), "last kernel before: " + prev["Kernel_Name"][:50], int(prev["End_Timestamp"]) - int(prev["Start_Timestamp"])))
    ev.append((int(k[i]["Start_Timestamp"]), "FIRST kernel of a volume", 0))
for r in m:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > t0 - 50_000_000: ev.append((s, f"copy {r.get('Direction', r.get('Kind', '?'))} {(e - s) / 1e6:.2f} ms", e - s))
ev.sort()
for t, what, d in ev[:120]: print(f"{(t - t0) / 1e6:10.3f} ms  {what}")
PY
