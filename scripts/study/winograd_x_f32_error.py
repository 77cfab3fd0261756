"""Would the x axis in Winograd F(2,3) form keep the exact-fp32 path inside its gate (<= 1.2 x the reference fp32 run's own distance from float64)?
One layer's worth of arithmetic in 1-D: out[x] = sum_k sum_dx d[k, x + dx] g[k, dx], K = 9 (dz, dy) x Cin, emulated in float32 with the kernels' summation
orders: `chain` = one running fp32 sum (what a plain fp32 conv does: ATen's order differs but has the same length), `two_level` = conv3_igemm_f32
(a fresh partial sum per 8-channel chunk of 27 taps, folded by one add), `wino_two_level` = conv3_wino_f32, F(2,3) along x: t = (d0 - d2, d1 + d2, d2 - d1,
d1 - d3) in fp32, u = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2) formed in double and rounded once, four sums over K (a fresh partial sum per
8-channel chunk of 9 taps), out = (m0 + m1 + m2, m1 - m2 - m3).  Prints rms error against float64 over the output rms.
(tests/test_oracle_golden.py::test_winograd_x_f32_emulation keeps the smallest case as a CPU test.)"""
import numpy as np


def seq_sum(terms, chunk):
    """terms [n, X] float32 summed sequentially in fp32: a fresh partial sum per `chunk` terms, folded into the running sum by one add"""
    acc = np.zeros(terms.shape[1], np.float32)
    for c0 in range(0, terms.shape[0], chunk):
        part = np.zeros(terms.shape[1], np.float32)
        for t in terms[c0:c0 + chunk]:
            part = (part + t).astype(np.float32)
        acc = (acc + part).astype(np.float32)
    return acc


def errors(cin, X=4096, seed=0):
    """(one chain, two-level direct, two-level Winograd x): rms error against float64 over the output rms"""
    rng = np.random.default_rng(seed)
    K = 9 * cin
    d = np.maximum(rng.normal(0.3, 1.0, (K, X + 2)), 0).astype(np.float32)          # post-ReLU activations
    g = (rng.normal(0, 1, (K, 3)) * np.sqrt(2.0 / (27 * cin))).astype(np.float32)
    truth = sum(d[:, dx:dx + X].astype(np.float64) * g[:, dx:dx + 1].astype(np.float64) for dx in range(3)).sum(0)
    prods = np.concatenate([(d[:, dx:dx + X] * g[:, dx:dx + 1]).astype(np.float32) for dx in range(3)])          # exact products rounded to fp32
    prods = prods[np.argsort(np.tile(np.arange(K), 3), kind="stable")]                                            # k-major: (k, dx)
    chain = seq_sum(prods, 10 ** 9)
    two = seq_sum(prods, 27 * 8)                                                                                     # 8 channels x 27 taps (here: 72 k x 3 dx)
    d4 = np.concatenate([d, np.zeros((K, 1), np.float32)], 1)
    d0, d1, d2, d3 = (d4[:, i:i + X:2] for i in range(4))
    tf = [(d0 - d2).astype(np.float32), (d1 + d2).astype(np.float32), (d2 - d1).astype(np.float32), (d1 - d3).astype(np.float32)]
    g64 = g.astype(np.float64)
    uf = [g64[:, 0], (g64[:, 0] + g64[:, 1] + g64[:, 2]) / 2, (g64[:, 0] - g64[:, 1] + g64[:, 2]) / 2, g64[:, 2]]
    uf = [u.astype(np.float32)[:, None] for u in uf]
    m = [seq_sum((tf[f] * uf[f]).astype(np.float32), 9 * 8) for f in range(4)]
    w0 = ((m[0] + m[1]).astype(np.float32) + m[2]).astype(np.float32)
    w1 = ((m[1] - m[2]).astype(np.float32) - m[3]).astype(np.float32)
    wino = np.empty(X, np.float32)
    wino[0::2] = w0
    wino[1::2] = w1
    sl = slice(0, X - 2)                      # (the last pair reads d[X + 2] = 0 instead of nothing)
    rms = np.sqrt((truth[sl] ** 2).mean())
    e = lambda v: float(np.sqrt(((v[sl].astype(np.float64) - truth[sl]) ** 2).mean()) / rms)
    return e(chain), e(two), e(wino)


if __name__ == "__main__":
    for cin in (64, 192, 768):
        c, t, w = errors(cin)
        print(f"Cin {cin:4d} (K = {9 * cin}): one chain {c:.2e}   two-level direct {t:.2e}   two-level Winograd x {w:.2e}   ratio wino / two-level {w / t:.2f}, wino / chain {w / c:.2f}")
