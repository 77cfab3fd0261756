"""Numerical study (CPU, numpy): would the y axis in Winograd F(2,3) form as well (2 x 2 outputs from 4 x 4 inputs: 16 products per (dz, cin) instead of 36;
12 MFMA-taps per output where the x-only form has 18 and the direct form 27) keep the fp16x3 conv at fp32-conv level?  Same set-up as winograd_x_error.py:
one layer Conv3d(Cin -> Cout, k3 p1) on ReLU-like activations calibrated to the [2^10, 2^11) window, He-scaled weights with per-cout power-of-two scales,
truth = float64; errors relative to the rms of the output.  The transformed inputs t = B^T d B are sums of FOUR 22-bit values, formed in fp32 and split into
an fp16 pair again (as the kernel would); the transformed weights are made in float64 and rounded once."""
import numpy as np
rng = np.random.default_rng(0)

def split(x):
    h0 = x.astype(np.float16).astype(np.float32)
    h1 = (x - h0).astype(np.float16).astype(np.float32)
    return h0, h1

def mm3(a, b):
    a0, a1 = split(a); b0, b1 = split(b)
    return (a0 @ b0 + a0 @ b1) + a1 @ b0

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)

for cin, cout in ((32, 32), (64, 64), (128, 128), (192, 64)):
    Z, Y, X = 6, 18, 34
    act = np.maximum(rng.standard_normal((Z, Y, X, cin)), 0).astype(np.float32)
    act *= np.float32(2.0 ** 10.5 / act.max())
    act = sum(split(act))
    w = (rng.standard_normal((3, 3, 3, cin, cout)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    ws = 2.0 ** (8 - np.ceil(np.log2(np.abs(w).reshape(-1, cout).max(0))))
    wsc = (w * ws).astype(np.float32)
    oz, oy, ox = Z - 2, Y - 2, X - 2
    cols = np.stack([act[dz:dz + oz, dy:dy + oy, dx:dx + ox] for dz in range(3) for dy in range(3) for dx in range(3)], axis=3).reshape(oz * oy * ox, 27 * cin)
    truth = cols.astype(np.float64) @ w.reshape(27 * cin, cout).astype(np.float64)
    rms = np.sqrt((truth ** 2).mean())
    direct3 = mm3(cols, wsc.reshape(27 * cin, cout)) / ws
    f32 = cols @ w.reshape(27 * cin, cout)
    res = [("fp32 direct", f32), ("fp16x3 direct", direct3)]
    # x only (the shipped form)
    P = ox // 2
    d = [act[:, :, k:k + 2 * P:2] for k in range(4)]
    t = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]
    g = wsc.astype(np.float64)
    u = [g[:, :, 0], (g[:, :, 0] + g[:, :, 1] + g[:, :, 2]) / 2, (g[:, :, 0] - g[:, :, 1] + g[:, :, 2]) / 2, g[:, :, 2]]
    m = []
    for f in range(4):
        colsf = np.stack([t[f][dz:dz + oz, dy:dy + oy] for dz in range(3) for dy in range(3)], axis=3).reshape(oz * oy * P, 9 * cin)
        m.append(mm3(colsf.astype(np.float32), u[f].reshape(9 * cin, cout).astype(np.float32)))
    y0 = (m[0] + m[1]) + m[2]; y1 = (m[1] - m[2]) - m[3]
    res.append(("fp16x3 winograd-x", np.stack([y0, y1], axis=1).reshape(oz * oy, P, 2, cout).reshape(oz * oy * ox, cout) / ws))
    # x and y: tiles of 2 x 2 outputs from 4 x 4 inputs
    Q = oy // 2
    D = np.stack([np.stack([act[:, ky:ky + 2 * Q:2, kx:kx + 2 * P:2] for kx in range(4)], axis=0) for ky in range(4)], axis=0)      # (4 ky, 4 kx, Z, Q, P, cin) fp32
    # t = B^T D B in fp32, the order the kernel would use: y first (adds of exact 22-bit values), then x
    ty = [D[0] - D[2], D[1] + D[2], D[2] - D[1], D[1] - D[3]]                                                                      # each (4 kx, Z, Q, P, cin)
    T = [[(a[0] - a[2]), (a[1] + a[2]), (a[2] - a[1]), (a[1] - a[3])] for a in ty]                                                  # T[fy][fx]: (Z, Q, P, cin), fp32
    U = np.einsum("ay,zyxio,bx->abzio", G, g, G)                                                                                  # (4, 4, 3 dz, cin, cout) float64
    M = np.zeros((4, 4, oz * Q * P, cout), dtype=np.float32)
    for fy in range(4):
        for fx in range(4):
            colsf = np.stack([T[fy][fx][dz:dz + oz] for dz in range(3)], axis=3).reshape(oz * Q * P, 3 * cin)
            M[fy, fx] = mm3(colsf.astype(np.float32), U[fy, fx].reshape(3 * cin, cout).astype(np.float32))
    # output transform in fp32: A^T M A (y then x)
    my = [(M[0] + M[1]) + M[2], (M[1] - M[2]) - M[3]]                                                                             # each (4 fx, rows, cout)
    out = np.zeros((oz, Q, 2, P, 2, cout), dtype=np.float32)
    for oyy in range(2):
        a = my[oyy]
        o0 = (a[0] + a[1]) + a[2]; o1 = (a[1] - a[2]) - a[3]
        out[:, :, oyy, :, 0] = o0.reshape(oz, Q, P, cout); out[:, :, oyy, :, 1] = o1.reshape(oz, Q, P, cout)
    res.append(("fp16x3 winograd-xy", out.reshape(oz * oy * ox, cout) / ws))
    for name, v in res:
        e = v.astype(np.float64) - truth
        print(f"Cin {cin:3d} Cout {cout:3d}  {name:18s} max|err|/rms {np.abs(e).max() / rms:.2e}   rms err / rms {np.sqrt((e ** 2).mean()) / rms:.2e}")
