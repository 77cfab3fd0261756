#!/bin/bash
# Diagnostic: time of ec1 (the first conv3_igemm_sres launch of a pass) under the OAI_DBG timing switches, with ec0 as its own launch
# (fuse_first=0) and fused into ec1's halo staging (fuse_first=1).  Needs the DIAGNOSTIC library; results are wrong when OAI_DBG != 0.
# bits: 1 no halo DMA, 4 no ec0 FMAs (fused), 16 no copy-out stores, 32 no fused pool stores
export OAI_LIB_PATH=${OAI_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/build/diag/liboai_hip_diag.so}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dbg_ec1; mkdir -p $O; cd $R
export PREC=fp16x3
for f in 0 1; do for d in 0 1 4 16 48 52 53; do
  export OAI_DBG=$d OPTIONS=fuse_first=$f
  rocprofv3 --kernel-trace -d $O/f${f}d$d -o t --output-format csv -- python3 scripts/perf_layers.py > $O/f${f}d$d.log 2>&1
done; done
python3 - <<'PY'
import csv, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/dbg_ec1"
for f in (0, 1):
    out = []
    for d in (0, 1, 4, 16, 48, 52, 53):
        rows = list(csv.DictReader(open(f"{O}/f{f}d{d}/t_kernel_trace.csv")))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        conv = [r for r in rows if "conv3_igemm_sres" in r["Kernel_Name"]]
        per = len(conv) // 2
        ec1 = conv[per]                                   # first conv launch of the second pass
        first = [r for r in rows if "conv3_first" in r["Kernel_Name"]]
        t = (int(ec1["End_Timestamp"]) - int(ec1["Start_Timestamp"])) / 1e3
        t0 = (int(first[-1]["End_Timestamp"]) - int(first[-1]["Start_Timestamp"])) / 1e3 if first else 0.0
        out.append(f"dbg={d}: ec1 {t:.0f} us (+ec0 {t0:.0f})")
    print(f"fuse_first={f}: " + "; ".join(out))
PY
