#!/bin/bash
# Round profile: (1) rocprofv3 kernel trace + stats of the default bench command, (2) HBM-side traffic counters of the
# split-resident segmentation kernels in their own passes (counters + kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/bench -o bench --output-format csv -- python3 bench.py --steps 3 --warmup 1 > $O/bench_line.json 2> $O/bench.err
tail -1 $O/bench_line.json | head -c 3000; echo
export PREC=fp16x3
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 scripts/perf_layers.py > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 scripts/perf_layers.py > $O/write.log 2>&1
python3 - <<'PY'
import csv, os, json
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/round"
def load(path, name):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows
f = load(O + "/fetch/f_counter_collection.csv", "FETCH_SIZE"); w = load(O + "/write/w_counter_collection.csv", "WRITE_SIZE")
def warm(rows):                                       # second (warm) 32-tile pass = from the last first-conv launch on
    i = max(k for k, r in enumerate(rows) if "conv3_first" in r["Kernel_Name"])
    return rows[i:]
f = warm(f); w = warm(w)
lines = ["| kernel (dispatch order, warm 32-tile pass) | FETCH raw MiB | FETCH x2 MiB | WRITE MiB |", "|---|---|---|---|"]
tot = {"conv_f": 0.0, "conv_w": 0.0, "conv_n": 0, "all_f": 0.0, "all_w": 0.0}
for a, b in zip(f, w):
    fk, wk = float(a["Counter_Value"]) / 1024, float(b["Counter_Value"]) / 1024
    name = a["Kernel_Name"][:64]
    lines.append(f"| {name} | {fk:.0f} | {2*fk:.0f} | {wk:.0f} |")
    tot["all_f"] += 2 * fk; tot["all_w"] += wk
    if "conv3_igemm_sres" in name: tot["conv_f"] += 2 * fk; tot["conv_w"] += wk; tot["conv_n"] += 1
open(O + "/traffic_table.md", "w").write("\n".join(lines) + "\n")
js = {"kernel": "conv3_igemm_sres (all tile shapes)", "bytes_per_launch": (tot["conv_f"] + tot["conv_w"]) * 2**20 / max(tot["conv_n"], 1),
      "fetch_x2_bytes_per_32_tile_pass": tot["conv_f"] * 2**20, "write_bytes_per_32_tile_pass": tot["conv_w"] * 2**20,
      "launches_per_32_tile_pass": tot["conv_n"], "all_kernels_fetch_x2_bytes": tot["all_f"] * 2**20, "all_kernels_write_bytes": tot["all_w"] * 2**20}
json.dump(js, open(O + "/traffic_sres.json", "w"), indent=1)
print(json.dumps(js))
PY
