"""Per-layer table of one full-size exact-fp32 segmentation pass from a `rocprofv3 --kernel-trace` of `PREC=f32 scripts/trace_layers.py`
(conv3_wino_f32 by default, conv3_igemm_f32 with OPTIONS=winograd_f32=0).  usage: python scripts/per_layer_table_f32.py <kernel_trace.csv>
Algorithmic GFLOP per tile and layer: the trimmed figures of SURVEY.md appendix B.1, as scripts/per_layer_table.py."""
import csv, sys
GF = {"ec1": 58.0, "ec2": 14.5, "ec3": 29.0, "ec4": 7.25, "ec5": 14.5, "ec6": 3.62, "ec7": 7.25, "dc8": 76.4, "dc7": 22.2, "dc5": 86.1,
      "dc4": 22.1, "dc2": 114.7, "dc1": 32.6}
ORDER = ["ec1", "ec2", "ec3", "ec4", "ec5", "ec6", "ec7", "dc9 (up)", "dc8", "dc7", "dc6 (up)", "dc5", "dc4", "dc3 (up)", "dc2", "dc1"]
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[[i for i, r in enumerate(rows) if "conv3_first" in r["Kernel_Name"]][-1]:]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
layers, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    main = "conv3_wino_f32<8, 4" in n or "conv3_igemm_f32<2, 8, 16, 2, 4, 1>" in n
    if "conv3_first" in n: layers.append(["ec0", dur(r), 1, ""]); continue
    if "upconv2" in n or main: layers.append([None, dur(r), 1, "wino" if "wino" in n else "direct" if "igemm" in n else ""]); cur = layers[-1]; continue
    if ("conv3_wino_f32" in n or "conv3_igemm_f32" in n) and cur is not None: cur[1] += dur(r); cur[2] += 1; continue       # strips
    layers.append([n.split("(")[0][-40:], dur(r), 1, ""])
names = iter(ORDER)
print("| layer | kernel | launches | us (160 tiles) | algorithmic TFLOP (160 tiles) | TFLOP/s | of 157.3 TFLOP/s |")
print("|---|---|---|---|---|---|---|")
tot = 0.0
for L in layers:
    if L[0] is None: L[0] = next(names)
    tot += L[1]
    gf = GF.get(L[0])
    tfs = gf * 160 / 1e3 / (L[1] * 1e-6) if gf else 0.0
    if gf: print(f"| {L[0]} | {L[3]} | {L[2]} | {L[1]:.0f} | {gf * 160 / 1e3:.2f} | {tfs:.1f} | {tfs / 157.3:.3f} |")
    else: print(f"| {L[0]} | | {L[2]} | {L[1]:.0f} | | | |")
print(f"\nsum of kernel time: {tot / 1e3:.1f} ms")
