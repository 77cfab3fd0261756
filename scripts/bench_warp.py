"""BASELINE config 2: the ICON warp / compose loop at 160^3 (HBM-bound kernels), GB/s against the 8 TB/s peak.

Algorithmic bytes per output voxel: image warp 20 B (12 coords + 4 src + 4 out), compose 36 B same-res
(12 coords + 12 src + 12 out), 25.5 B from a half-resolution source.
"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd import ops
from oai_analysis_2_amd.synth import identity_map, make_smooth_field, make_volume

if os.environ.get("BRICK"):        # BRICK=1: the LDS-staged brick form of grid_sample3d / compose (option "brick", round 5)
    ops.warp_set_option("brick", int(os.environ["BRICK"]))
N = int(os.environ.get("N", "160"))
shape = (N, N, N)
V = N ** 3
img = torch.from_numpy(make_volume(1, shape))[None].cuda()
d_full = torch.from_numpy(make_smooth_field(2, shape, 0.02)).cuda()
d_half = torch.from_numpy(make_smooth_field(3, (N // 2,) * 3, 0.02)).cuda()
coords = (torch.from_numpy(identity_map(shape)) + torch.from_numpy(make_smooth_field(4, shape, 0.03))).cuda().contiguous()

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

res = {}
for name, fn, bytes_per_voxel in [
        ("grid_sample3d (image warp, K14)", lambda: ops.grid_sample3d(img, coords), 20.0),
        ("compose same-res (K15)", lambda: ops.compose(d_full, coords), 36.0),
        ("compose half-res source (K15)", lambda: ops.compose(d_half, coords), 25.5),
        ("compose identity coords, half-res source", lambda: ops.compose(d_half, None, out_shape=shape, shortcut=False), 13.5)]:
    t = timeit(fn)
    gbs = bytes_per_voxel * V / t / 1e9
    res[name] = {"us": t * 1e6, "GB/s": gbs, "frac_of_8TB/s": gbs / 8000}
    print(f"{name:45s} {t*1e6:8.1f} us  {gbs:8.1f} GB/s  {gbs/8000:.3f} of 8 TB/s")
print(json.dumps(res))

# ---- against HBM for real: one warp touches 82 MB (coords 49 + src 16 + out 16), less than the 256 MB Infinity Cache, so
# the repeated call above is a MALL-resident measurement.  Here K distinct (src, coords, out) sets are cycled (K x 82 MB > 256 MB for
# K >= 4; outputs are preallocated and written through the C ABI directly): every call finds its inputs in HBM.  ROT=K (default 8).
import ctypes as C
from oai_analysis_2_amd import _lib
lib = _lib.load()
K = int(os.environ.get("ROT", "8"))
sets = []
for k in range(K):
    s_img = torch.from_numpy(make_volume(10 + k, shape))[None].cuda()
    s_co = (torch.from_numpy(identity_map(shape)) + torch.from_numpy(make_smooth_field(20 + k, shape, 0.03))).cuda().contiguous()
    sets.append((s_img, s_co, torch.empty((1, N, N, N), dtype=torch.float32, device="cuda")))
st = torch.cuda.current_stream().cuda_stream
def warp_k(k):
    s_img, s_co, out = sets[k % K]
    _lib.check(lib.oai_grid_sample3d(s_img.data_ptr(), 1, N, N, N, s_co.data_ptr(), N, N, N, out.data_ptr(), st), "oai_grid_sample3d")
for mode, kk in (("one set repeated (Infinity-Cache resident)", 1), (f"{K} sets cycled = {K * 82} MB (HBM)", K)):
    for i in range(2 * K): warp_k(i % kk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 40 * K
    e0.record()
    for i in range(iters): warp_k(i % kk)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    gbs = 20.0 * V / t / 1e9
    res["grid_sample3d " + mode] = {"us": t * 1e6, "GB/s": gbs, "frac_of_8TB/s": gbs / 8000, "frac_of_6.29TB/s": gbs / 6290}
    print(f"grid_sample3d {mode:48s} {t*1e6:8.1f} us  {gbs:8.1f} GB/s  {gbs/8000:.3f} of 8 TB/s  {gbs/6290:.3f} of 6.29 TB/s")
# a pure streaming kernel of the same byte count over the same rotation, as the yardstick of what HBM delivers to ANY kernel here:
# torch's elementwise add over 5 x 16.4 MB tensors per call (2 reads + ... ) would be another product's kernel; use a copy of the coords
# (49 MB read + 49 MB written = 98 MB) through cudaMemcpyAsync D2D instead
dst = torch.empty_like(sets[0][1])
for mode, kk in ((f"D2D copy of the coords, {K} sets cycled", K),):
    for i in range(2 * K): dst.copy_(sets[i % kk][1])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 40 * K
    e0.record()
    for i in range(iters): dst.copy_(sets[i % kk][1])
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    gbs = 2 * 12.0 * V / t / 1e9
    res[mode] = {"us": t * 1e6, "GB/s": gbs}
    print(f"{mode:62s} {t*1e6:8.1f} us  {gbs:8.1f} GB/s (read + write)")
print(json.dumps(res))
