"""Timing of the mesh / thickness step on a full-size 160x384x384 probability map: a cartilage-like curved
slab (5 voxels thick, ~300 x 120 voxels wide), the sizes of mesh_processing.get_thickness_mesh on a real femoral cartilage."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd import mesh_processing as mp
from oai_analysis_2_amd.image import Image

D, H, W = 160, 384, 384
z, y, x = np.mgrid[0:D, 0:H, 0:W].astype(np.float32)
sig = lambda t: 1.0 / (1.0 + np.exp(np.clip(t, -60, 60)))
R, T = 220.0, 5.0
r = np.sqrt((x - 192) ** 2 + ((z - 80) * 1.9) ** 2 + (y + 60) ** 2)
prob = (sig(2.0 * (np.abs(r - R) - T / 2)) * sig(2.0 * (np.sqrt((x - 192) ** 2 + ((z - 80) * 1.9) ** 2) - 140))).astype(np.float32)
img = Image(prob, [0.36, 0.36, 0.7])
vol = torch.from_numpy(prob).cuda()

def timed(name, fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(reps): out = fn()
    torch.cuda.synchronize()
    dt = (time.time() - t) / reps
    print(f"{name:52s} {dt*1e3:9.2f} ms")
    return out

v, f = timed("marching cubes 160x384x384 (count + emit + D2H)", lambda: mp.marching_cubes(vol, 0.5, img.spacing))
print(f"   {len(v)} vertices, {len(f)} triangles; volume read 2x94 MB + 5 x 94 MB of counts/offsets")
v, f = mp.keep_large_regions(v, f, 3000)
mesh = mp.Mesh(v, f)
sm = timed("smooth 150 iterations (incl. host edge graph)", lambda: mp.smooth_mesh(mesh, 150))
t = time.time(); inner, outer = mp.split_mesh(sm, "FC"); print(f"{'split (host KMeans, as the reference)':52s} {(time.time()-t)*1e3:9.2f} ms   inner {len(inner.faces)} / outer {len(outer.faces)} faces")
d = timed("distance both directions (uniform-grid broad phase)", lambda: mp.get_distance(inner, outer), reps=2)
import functools
mp_point = mp.point_distance
timed("distance both directions (brute force)", lambda: (mp_point(inner.verts, outer, False), mp_point(outer.verts, inner, False)), reps=2)
print("   median thickness", float(np.median(d[0].point_data["Distance"])), "(slab: 5 voxels x 0.36-0.7 mm)")
pairs = len(inner.verts) * len(outer.faces) + len(outer.verts) * len(inner.faces)
print(f"   brute force = {pairs/1e9:.2f} G point-triangle tests")
