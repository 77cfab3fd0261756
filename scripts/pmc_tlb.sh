#!/bin/bash
# PMC: address-translation (UTCL1) and L1 stall counters of the conv kernel, with and without the halo DMA (OAI_DBG=1)
# needs the DIAGNOSTIC library (python -m oai_analysis_2_amd.build --diag): the production library ignores OAI_DBG
export OAI_LIB_PATH=${OAI_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/build/diag/liboai_hip_diag.so}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_tlb; mkdir -p $O; cd $R
export PREC=fp16x3
for d in 0 1; do
  export OAI_DBG=$d
  i=0
  for set in "TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_STALL_INFLIGHT_MAX" \
             "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_TOTAL_CACHE_ACCESSES" \
             "TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_STALL_LRU_INFLIGHT TCP_UTCL1_THRASHING_STALL" \
             "TCP_GATE_EN1 TCP_GATE_EN2 TCP_TCP_TA_DATA_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set -d $O/d${d}_p$i -o p --output-format csv -- python3 scripts/perf_layers.py > $O/d${d}_p$i.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_tlb"
for d in (0, 1):
    agg = collections.defaultdict(float)
    for f in glob.glob(f"{O}/d{d}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv3_igemm_sres<4, 16, 2, 4, 1" in r["Kernel_Name"]: agg[r["Counter_Name"]] += float(r["Counter_Value"])
    print("OAI_DBG =", d)
    for k, v in sorted(agg.items()): print(f"   {k:36s} {v:.4g}")
PY
