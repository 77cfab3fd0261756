"""Numerical study (CPU, numpy): error of an x-only Winograd F(2,3) form of the fp16x3 conv against the direct fp16x3 form and plain fp32.

A layer = Conv3d(Cin -> Cout, k3 p1) on ReLU-like activations calibrated to the [2^10, 2^11) window, He-scaled weights with a per-cout
power-of-two scale (as pack_fp16_layer does).  Truth = float64.  Reports max / rms error relative to the rms of the output.
"""
import numpy as np
rng = np.random.default_rng(0)

def split(x):                       # fp32 -> (h0, h1) fp16 pair, as float32 arrays
    h0 = x.astype(np.float16).astype(np.float32)
    h1 = (x - h0).astype(np.float16).astype(np.float32)
    return h0, h1

def mm3(a, b):                      # three-pass product with fp32 accumulation: a0 b0 + a0 b1 + a1 b0
    a0, a1 = split(a); b0, b1 = split(b)
    return (a0 @ b0 + a0 @ b1) + a1 @ b0

for cin, cout in ((32, 32), (64, 64), (128, 128)):
    Z, Y, X = 6, 10, 34
    act = np.maximum(rng.standard_normal((Z, Y, X, cin)), 0).astype(np.float32)
    act *= np.float32(2.0 ** 10.5 / act.max())
    act = sum(split(act))                                        # what format S holds
    w = (rng.standard_normal((3, 3, 3, cin, cout)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    ws = 2.0 ** (8 - np.ceil(np.log2(np.abs(w).reshape(-1, cout).max(0))))          # per-cout power of two
    wsc = (w * ws).astype(np.float32)
    oz, oy, ox = Z - 2, Y - 2, X - 2
    # im2col rows for the direct form
    cols = np.stack([act[dz:dz + oz, dy:dy + oy, dx:dx + ox] for dz in range(3) for dy in range(3) for dx in range(3)], axis=3).reshape(oz * oy * ox, 27 * cin)
    truth = cols.astype(np.float64) @ w.reshape(27 * cin, cout).astype(np.float64)
    rms = np.sqrt((truth ** 2).mean())
    direct3 = mm3(cols, wsc.reshape(27 * cin, cout)) / ws
    f32 = cols @ w.reshape(27 * cin, cout)
    # Winograd F(2,3) along x: outputs (2p, 2p+1) from inputs d0..d3 = act[..., 2p : 2p+4]
    P = ox // 2
    d = [act[:, :, k:k + 2 * P:2] for k in range(4)]             # each (Z, Y, P, cin), fp32 adds of exact 22-bit values
    t = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]
    g = wsc.astype(np.float64)
    u = [g[:, :, 0], (g[:, :, 0] + g[:, :, 1] + g[:, :, 2]) / 2, (g[:, :, 0] - g[:, :, 1] + g[:, :, 2]) / 2, g[:, :, 2]]     # (3,3,cin,cout) each, in float64 then rounded once
    m = []
    for f in range(4):
        colsf = np.stack([t[f][dz:dz + oz, dy:dy + oy] for dz in range(3) for dy in range(3)], axis=3).reshape(oz * oy * P, 9 * cin)
        m.append(mm3(colsf.astype(np.float32), u[f].reshape(9 * cin, cout).astype(np.float32)))
    y0 = (m[0] + m[1]) + m[2]
    y1 = (m[1] - m[2]) - m[3]
    wino = np.stack([y0, y1], axis=1).reshape(oz * oy, P, 2, cout).reshape(oz * oy * ox, cout) / ws
    for name, v in (("fp32 direct", f32), ("fp16x3 direct", direct3), ("fp16x3 winograd-x", wino)):
        e = v.astype(np.float64) - truth
        print(f"Cin {cin:3d} Cout {cout:3d}  {name:18s} max|err|/rms {np.abs(e).max() / rms:.2e}   rms err / rms {np.sqrt((e ** 2).mean()) / rms:.2e}")
