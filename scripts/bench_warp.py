"""BASELINE config 2: the ICON warp / compose loop at 160^3 (HBM-bound kernels), GB/s against the 8 TB/s peak.

Algorithmic bytes per output voxel: image warp 20 B (12 coords + 4 src + 4 out), compose 36 B same-res
(12 coords + 12 src + 12 out), 25.5 B from a half-resolution source.
"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd import ops
from oai_analysis_2_amd.synth import identity_map, make_smooth_field, make_volume

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
