#!/bin/bash
# Diagnostic: per-launch time of the split-resident conv kernel with runtime switches (OAI_DBG bits; results wrong when non-zero).
# needs the DIAGNOSTIC library (python -m oai_analysis_2_amd.build --diag): the production library ignores OAI_DBG
export OAI_LIB_PATH=${OAI_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/build/diag/liboai_hip_diag.so}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dbg; mkdir -p $O; cd $R
export PREC=fp16x3
for d in 0 1 2 16 48 51; do
  export OAI_DBG=$d
  rocprofv3 --kernel-trace -d $O/d$d -o d$d --output-format csv -- python3 scripts/perf_layers.py > $O/d$d.log 2>&1
done
python3 - <<'PY'
import csv, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/dbg"
cols = {}
for d in (0, 1, 2, 16, 48, 51):
    rows = [r for r in csv.DictReader(open(f"{O}/d{d}/d{d}_kernel_trace.csv")) if "igemm_sres" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    rows = rows[len(rows) // 2:]            # second repetition
    cols[d] = [(r["Kernel_Name"][22:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
n = len(cols[0])
print("launch                                   " + "".join(f"dbg={d:<6d}" for d in cols))
for i in range(n):
    print(f"{i:2d} {cols[0][i][0]:38s}" + "".join(f"{cols[d][i][1]:9.0f} " for d in cols))
print("sum" + " " * 38 + "".join(f"{sum(x[1] for x in cols[d]):9.0f} " for d in cols))
PY
