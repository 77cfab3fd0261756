#!/bin/bash
# Round profile: (1) rocprofv3 kernel trace + stats of the default bench command, (2) HBM-side traffic counters of the
# split-resident segmentation kernels in their own passes (counters + kernel-trace only), (3) SQ / GRBM counters of the dominant
# kernel (MFMA-busy share, effective clock = GRBM_GUI_ACTIVE / 8 / wall).   usage: bash scripts/prof_round.sh [round tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}; O=$R/gpurun_out/round_$TAG; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/bench -o bench --output-format csv -- python3 bench.py --steps 3 --warmup 1 > $O/bench_line.json 2> $O/bench.err
tail -1 $O/bench_line.json | head -c 1500; echo
export PREC=f32            # the exact-fp32 leg's HBM-side traffic (its own passes; counters + kernel-trace only)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch32 -o f --output-format csv -- python3 scripts/perf_layers.py > $O/fetch32.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write32 -o w --output-format csv -- python3 scripts/perf_layers.py > $O/write32.log 2>&1
export PREC=fp16x3
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 scripts/perf_layers.py > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 scripts/perf_layers.py > $O/write.log 2>&1
rocprofv3 --kernel-trace -d $O/layers -o t --output-format csv -- python3 scripts/trace_layers.py > $O/layers.log 2>&1
python3 scripts/per_layer_table.py $(find $O/layers -name "*kernel_trace.csv" | head -1) > $O/conv_per_layer_table.md
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $O/sq -o s --output-format csv -- python3 scripts/perf_layers.py > $O/sq.log 2>&1
python3 - "$O" <<'PY'
import csv, os, json, sys, glob, collections
O = sys.argv[1]
def find(d, suffix):
    fs = glob.glob(os.path.join(O, d, "**", "*" + suffix), recursive=True)
    return fs[0] if fs else None
def load(path, name):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows
def traffic_f32():
    """exact-fp32 leg: bytes per k3 launch (conv3_wino_f32, round 6; conv3_igemm_f32 before / with winograd_f32=0) of the warm pass (no ec0 fusion there: a pass starts at conv3_first_kernel)"""
    f = load(find("fetch32", "counter_collection.csv"), "FETCH_SIZE"); w = load(find("write32", "counter_collection.csv"), "WRITE_SIZE")
    cut = lambda rows: rows[max(k for k, r in enumerate(rows) if "conv3_first" in r["Kernel_Name"]):]
    f, w = cut(f), cut(w)
    k3 = lambda a: "conv3_igemm_f32" in a["Kernel_Name"] or "conv3_wino_f32" in a["Kernel_Name"]
    tf = sum(2 * float(a["Counter_Value"]) * 1024 for a in f if k3(a))
    tw = sum(float(a["Counter_Value"]) * 1024 for a in w if k3(a))
    n = sum(1 for a in f if k3(a))
    js = {"kernel": "conv3_wino_f32 / conv3_igemm_f32 (all block shapes; main launches and strips count as launches, as in bench.py)", "bytes_per_launch": (tf + tw) / max(n, 1), "tiles_per_pass": int(os.environ.get("TILES", "160")),
          "fetch_x2_bytes_per_pass": tf, "write_bytes_per_pass": tw, "launches_per_pass": n}
    json.dump(js, open(O + "/traffic_f32.json", "w"), indent=1)
    print(json.dumps(js))
try: traffic_f32()
except Exception as e: print("f32 traffic:", e)
f = load(find("fetch", "counter_collection.csv"), "FETCH_SIZE"); w = load(find("write", "counter_collection.csv"), "WRITE_SIZE")
def warm(rows):                                       # second (warm) 32-tile pass = from its first launch on: ec0, or the ec0-fused ec1 (<..., false, true, false>)
    def is_first(name):                                # conv3_igemm_sres<MREP, RX, RY, WY, WX, RING, FIRST, BLDS, M16> with FIRST = true
        if "conv3_igemm_sres<" not in name: return False
        a = [x.strip() for x in name.split("conv3_igemm_sres<", 1)[1].split(">", 1)[0].split(",")]
        return len(a) > 6 and a[6] == "true"
    first = [k for k, r in enumerate(rows) if is_first(r["Kernel_Name"])] or \
            [k for k, r in enumerate(rows) if "conv3_first" in r["Kernel_Name"]]
    return rows[max(first):]
f = warm(f); w = warm(w)
lines = ["| kernel (dispatch order, warm full-volume pass) | FETCH raw MiB | FETCH x2 MiB | WRITE MiB |", "|---|---|---|---|"]
tot = {"conv_f": 0.0, "conv_w": 0.0, "conv_n": 0, "all_f": 0.0, "all_w": 0.0}
for a, b in zip(f, w):
    fk, wk = float(a["Counter_Value"]) / 1024, float(b["Counter_Value"]) / 1024
    name = a["Kernel_Name"][:64]
    lines.append(f"| {name} | {fk:.0f} | {2*fk:.0f} | {wk:.0f} |")
    tot["all_f"] += 2 * fk; tot["all_w"] += wk
    if "conv3_igemm_sres" in name or "conv3_wino_sres" in name: tot["conv_f"] += 2 * fk; tot["conv_w"] += wk; tot["conv_n"] += 1
open(O + "/traffic_table.md", "w").write("\n".join(lines) + "\n")
js = {"kernel": "conv3_igemm_sres / conv3_igemm_sres2 / conv3_wino_sres (all tile shapes)", "bytes_per_launch": (tot["conv_f"] + tot["conv_w"]) * 2**20 / max(tot["conv_n"], 1),
      "tiles_per_pass": int(os.environ.get("TILES", "160")),
      "fetch_x2_bytes_per_pass": tot["conv_f"] * 2**20, "write_bytes_per_pass": tot["conv_w"] * 2**20,
      "launches_per_pass": tot["conv_n"], "all_kernels_fetch_x2_bytes": tot["all_f"] * 2**20, "all_kernels_write_bytes": tot["all_w"] * 2**20}
json.dump(js, open(O + "/traffic_sres.json", "w"), indent=1)
print(json.dumps(js))
# SQ / GRBM sums per kernel + the kernel-trace durations of the same pass -> MFMA-busy share and effective clock
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(find("sq", "counter_collection.csv"))):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
dur = collections.defaultdict(float)
kt = find("sq", "kernel_trace.csv")
if kt:
    for r in csv.DictReader(open(kt)):
        dur[r["Kernel_Name"][:60]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
with open(O + "/sq_summary.md", "w") as fo:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:9]:
        t = dur.get(k, 0.0)
        clk = v.get("GRBM_GUI_ACTIVE", 0) / 8 / t / 1e9 if t else 0
        simd_cycles = clk * 1e9 * t * 1024
        fo.write(f"{k}\n  wall {t*1e3:.2f} ms, effective clock {clk:.3f} GHz, MFMA-busy / (1024 SIMDs x wall x clock) = "
                 f"{v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd_cycles if simd_cycles else 0:.3f}\n")
        for c, x in sorted(v.items()):
            fo.write(f"    {c:28s} {x:.4g}\n")
print(open(O + "/sq_summary.md").read()[:3000])
PY
