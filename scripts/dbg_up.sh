#!/bin/bash
# Diagnostic: up-conv kernel time with runtime switches (OAI_DBG 64 = no copy-out stores, 128 = no LDS reads/MFMAs, 256 = no DMA after the prologue, 512 = no epilogue image build)
# needs the DIAGNOSTIC library (python -m oai_analysis_2_amd.build --diag): the production library ignores OAI_DBG
export OAI_LIB_PATH=${OAI_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/build/diag/liboai_hip_diag.so}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dbgup; mkdir -p $O; cd $R
export PREC=fp16x3
for d in 0 64 128 256 512 576 448 960; do
  export OAI_DBG=$d
  rocprofv3 --kernel-trace -d $O/d$d -o d$d --output-format csv -- python3 scripts/perf_layers.py > $O/d$d.log 2>&1
done
python3 - <<'PY'
import csv, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/dbgup"
for d in (0, 64, 128, 256, 512, 576, 448, 960):
    rows = [r for r in csv.DictReader(open(f"{O}/d{d}/d{d}_kernel_trace.csv")) if "upconv2" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    rows = rows[len(rows) // 2:]
    print(f"dbg={d:4d} " + " ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.0f}" for r in rows))
PY
