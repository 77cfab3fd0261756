import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from oai_analysis_2_amd import mesh_processing as mp
from oai_analysis_2_amd.synth import make_volume
vol = torch.from_numpy(make_volume(3)).cuda()          # noisy 160x384x384 volume: a worst case for surface size
t = time.time(); v, f = mp.marching_cubes(vol, float(vol.median()), (0.36, 0.36, 0.7)); print("mc", time.time() - t, v.shape, f.shape)
assert f.max() < len(v) and f.min() >= 0
e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).astype(np.int64)
key = e[:, 0] * len(v) + e[:, 1]
rk = e[:, 1] * len(v) + e[:, 0]
ks = np.sort(key)
print("directed edges unique:", bool((np.diff(ks) > 0).all()))
pos = np.searchsorted(ks, rk); pos[pos >= len(ks)] = 0
has_rev = ks[pos] == rk
print("edges with a reverse (closed except at the volume border):", has_rev.mean())
