"""MFMA-busy share and effective clock per kernel from a `rocprofv3 --kernel-trace --pmc SQ_... GRBM_GUI_ACTIVE` directory.
usage: python scripts/sq_summary.py <dir> [max kernels]"""
import collections, csv, glob, os, sys
O = sys.argv[1]; N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
def find(suffix):
    fs = glob.glob(os.path.join(O, "**", "*" + suffix), recursive=True)
    return fs[0] if fs else None
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(find("counter_collection.csv"))):
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
dur, calls = collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(find("kernel_trace.csv"))):
    dur[r["Kernel_Name"][:70]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    calls[r["Kernel_Name"][:70]] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:N]:
    t = dur.get(k, 0.0)
    clk = v.get("GRBM_GUI_ACTIVE", 0) / 8 / t / 1e9 if t else 0
    simd_cycles = clk * 1e9 * t * 1024
    print(f"{k}\n  {calls[k]} launches, wall {t*1e3:.2f} ms, effective clock {clk:.3f} GHz, MFMA-busy / (1024 SIMDs x wall x clock) = "
          f"{v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd_cycles if simd_cycles else 0:.3f}")
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {x:.4g}")
