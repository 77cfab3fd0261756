"""Per-layer table of one full-size fp16x3 segmentation pass from a `rocprofv3 --kernel-trace` of scripts/trace_layers.py.
usage: python scripts/per_layer_table.py gpurun_out/layers/t_kernel_trace.csv > profiles/<tag>_conv_per_layer.md
Algorithmic GFLOP per tile and layer are the trimmed figures of SURVEY.md appendix B.1 (they sum to 505.4)."""
import csv, sys
GF = {"ec1": 58.0, "ec2": 14.5, "ec3": 29.0, "ec4": 7.25, "ec5": 14.5, "ec6": 3.62, "ec7": 7.25, "dc8": 76.4, "dc7": 22.2, "dc5": 86.1,
      "dc4": 22.1, "dc2": 114.7, "dc1": 32.6}
ORDER = ["ec1", "ec2", "ec3", "ec4", "ec5", "ec6", "ec7", "dc9 (up)", "dc8", "dc7", "dc6 (up)", "dc5", "dc4", "dc3 (up)", "dc2", "dc1"]
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def _targs(name):                                      # template arguments of conv3_igemm_sres<MREP, RX, RY, WY, WX, RING, FIRST, BLDS, M16>
    if "conv3_igemm_sres<" not in name: return []
    return [a.strip() for a in name.split("conv3_igemm_sres<", 1)[1].split(">", 1)[0].split(",")]
fused = [i for i, r in enumerate(rows) if len(_targs(r["Kernel_Name"])) > 6 and _targs(r["Kernel_Name"])[6] == "true"]
if fused: rows = rows[fused[-1]:]                        # ec0 fused into ec1 (per tile, or the shared pass over the padded volume): a pass starts at the FIRST instantiation
else:
    starts = [i for i, r in enumerate(rows) if "conv3_first" in r["Kernel_Name"]]
    rows = rows[starts[-1]:]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
layers, cur = [], None
# shared encoder pass (round 3): the FIRST launch over the padded volume is followed by pooled_gather, the ec0 shell, six face launches and the
# pooled faces -- all of it is "ec1"
if len(rows) > 1 and "pooled_gather" in rows[1]["Kernel_Name"]:
    k = next(i for i, r in enumerate(rows) if "maxpool2_sres" in r["Kernel_Name"])
    layers.append([None, sum(dur(r) for r in rows[:k + 1]), k + 1]); cur = layers[-1]
    rows = rows[k + 1:]
for r in rows:
    n = r["Kernel_Name"]
    main = "conv3_igemm_sres<4, 16, 2, 4, 1" in n or "conv3_igemm_sres2<16, 2, 4, 1" in n or ("conv3_wino_sres<" in n and ", 8, 4" in n)
    if "conv3_first" in n and not fused: layers.append(["ec0", dur(r), 1]); continue
    if "upconv2" in n or main: layers.append([None, dur(r), 1]); cur = layers[-1]; continue
    if ("conv3_igemm_sres" in n or "conv3_wino_sres" in n) and cur is not None: cur[1] += dur(r); cur[2] += 1; continue      # strip launches belong to the layer before them
    layers.append([n[:40], dur(r), 1])
names = iter(ORDER)
print("| layer | launches | us (160 tiles) | algorithmic TFLOP (160 tiles) | TFLOP/s | of 2.5 PFLOP/s |")
print("|---|---|---|---|---|---|")
tot = 0.0
for L in layers:
    if L[0] is None: L[0] = next(names)
    tot += L[1]
    gf = GF.get(L[0])
    tfs = gf * 160 / 1e3 / (L[1] * 1e-6) if gf else 0.0           # TFLOP / s
    if gf: print(f"| {L[0]} | {L[2]} | {L[1]:.0f} | {gf * 160 / 1e3:.2f} | {tfs:.0f} | {tfs / 2500:.3f} |")
    else: print(f"| {L[0]} | {L[2]} | {L[1]:.0f} | | | |")
print(f"\nsum of kernel time: {tot / 1e3:.1f} ms")
