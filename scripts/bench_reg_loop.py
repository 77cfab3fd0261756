"""BASELINE config 3: the ICON warp / compose loop at 160^3 (fields d1, d2 @80^3 and d3 @160^3 given, no nets), op by op and as
fused chains (oai_warp_chain), plus the fused two-map phi-resample at 384x384x160 and one ICON direction with / without graph replay.
Algorithmic bytes: SURVEY.md 8(d) -- op-by-op ~186 B/voxel, fused floor ~48 B/voxel; per chain as implemented:
  A: a(id_l + d1) at 80^3            (12 + 4 + 4) x V/8                    =  2.5 B/voxel of the 160^3 grid
  B: A(c2), c2 = c1 + d1(c1), c1 = id + d2(id)   2 x 12 x V/8 + 4 + 4      = 11   B/voxel
  C: phi = c4 + d1(c4), c4 = c3 + d2(c3), c3 = id + d3   12 + 3 + 12        = 27   B/voxel
  2 x avg_pool                                              2 x 4.5         =  9   B/voxel          total 49.5 B/voxel
resample (2 maps, straight from phi): 2 x (4 + 4) B per atlas voxel + 12 B per network voxel = 413 MB at 384x384x160 / 80x192x192."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd import ops
from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.registration import IconEngine, resample_affines
from oai_analysis_2_amd.synth import make_icon_state_dict, make_smooth_field, make_volume

def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

N = 160
hi, lo = (N,) * 3, (N // 2,) * 3
V = N ** 3
A = torch.from_numpy(make_volume(1, hi)).cuda()
B = torch.from_numpy(make_volume(2, hi)).cuda()
d1, d2 = (torch.from_numpy(make_smooth_field(s, lo, 0.02)).cuda() for s in (3, 4))
d3 = torch.from_numpy(make_smooth_field(5, hi, 0.02)).cuda()

def loop_op_by_op():
    a, b = ops.avgpool2(A[None])[0], ops.avgpool2(B[None])[0]
    aw = ops.grid_sample3d(a[None], ops.compose(d1, None, shortcut=True))
    c2 = ops.compose(d1, ops.compose(d2, None, out_shape=hi, shortcut=False))
    Aw = ops.grid_sample3d(A[None], c2)
    return ops.compose(d1, ops.compose(d2, ops.compose(d3, None, shortcut=True)))

def loop_fused():
    a, b = ops.avgpool2(A[None])[0], ops.avgpool2(B[None])[0]
    aw = ops.warp_chain(lo, start=d1, image=a)
    Aw = ops.warp_chain(hi, fields=[d2, d1], image=A)
    return ops.warp_chain(hi, fields=[d2, d1], start=d3)

assert torch.equal(loop_op_by_op(), loop_fused())
res = {}
t = timeit(loop_op_by_op); res["config-3 loop, op by op (186 B/voxel)"] = {"us": t * 1e6, "GB/s": 186 * V / t / 1e9, "frac_of_8TB/s": 186 * V / t / 8e12}
t = timeit(loop_fused); res["config-3 loop, fused chains (49.5 B/voxel)"] = {"us": t * 1e6, "GB/s": 49.5 * V / t / 1e9, "frac_of_8TB/s": 49.5 * V / t / 8e12}
for name, fn, bpv in [("chain B: A(c2) from d2, d1 (11 B/voxel)", lambda: ops.warp_chain(hi, fields=[d2, d1], image=A), 11.0),
                      ("chain C: phi from d3, d2, d1 (27 B/voxel)", lambda: ops.warp_chain(hi, fields=[d2, d1], start=d3), 27.0)]:
    t = timeit(fn); res[name] = {"us": t * 1e6, "GB/s": bpv * V / t / 1e9, "frac_of_8TB/s": bpv * V / t / 8e12}

shape, net = (160, 384, 384), (80, 192, 192)
maps = torch.stack([torch.from_numpy(make_volume(7, shape)), torch.from_numpy(make_volume(8, shape))]).cuda()
img = Image(make_volume(7, shape), [0.36, 0.36, 0.7], [3.0, -2.0, 1.0])
atlas = Image(make_volume(9, shape), [0.36, 0.36, 0.7], [0.0, 0.0, 0.0])
ident = torch.from_numpy(np.stack(np.meshgrid(*[np.arange(n, dtype=np.float64) / (n - 1) for n in net], indexing="ij")).astype(np.float32)).cuda()
phi = (ident + torch.from_numpy(make_smooth_field(11, net, 0.02)).cuda()).contiguous()
b2n, n2a = resample_affines(img, atlas, net)
nb = 2 * 8 * maps[0].numel() + 12 * phi[0].numel()
t = timeit(lambda: ops.resample_maps_through_phi(maps, phi, b2n, n2a, shape))
res["resample 2 maps through phi, fused K18+K19 (413 MB)"] = {"us": t * 1e6, "GB/s": nb / t / 1e9, "frac_of_8TB/s": nb / t / 8e12}
disp = ops.phi_to_itk_displacement(phi)
t = timeit(lambda: [ops.resample_through_disp(maps[c], disp, b2n, n2a, shape) for c in range(2)])
res["resample 2 maps, round-1 path (fp64 displacement, one launch per map; 448 MB)"] = {"us": t * 1e6, "GB/s": 2 * 9.5 * maps[0].numel() / t / 1e9,
                                                                                     "frac_of_8TB/s": 2 * 9.5 * maps[0].numel() / t / 8e12}
eng = IconEngine(make_icon_state_dict(0, 0.05), net)
An, Bn = torch.from_numpy(make_volume(1, net)).cuda(), torch.from_numpy(make_volume(2, net)).cuda()
for g in (True, False):
    eng.set_graph(g)
    t = timeit(lambda: eng.phi(An, Bn), 10)
    res[f"ICON one direction 80x192x192, graph replay {'on' if g else 'off'}"] = {"us": t * 1e6}
for k, v in res.items():
    print(f"{k:85s} " + "  ".join(f"{a}={b:.1f}" if a != "frac_of_8TB/s" else f"{a}={b:.3f}" for a, b in v.items()))
print(json.dumps(res))
