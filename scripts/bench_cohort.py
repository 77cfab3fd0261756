"""BASELINE config 4: fused segment -> register -> resample per volume, N synthetic 384x384x160 volumes streamed through one MI355X from
HOST memory -- which part of the streaming costs what (round 5): resident loop (pipe.run on one device tensor), results left on the device
(upload only), results copied back (upload + D2H + host copy on worker threads).  Steady state = the inter-result interval away from fill / drain."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.cohort import CohortRunner
from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.pipeline import VolumePipeline
from oai_analysis_2_amd.registration import IconEngine
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume

N = int(os.environ.get("N", "16"))
shape = (160, 384, 384)
meta = dict(spacing=[0.36, 0.36, 0.7], origin=[0.0, 0.0, 0.0])
atlas = Image(make_volume(1000, shape), **meta)
unet = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
icon = IconEngine(make_icon_state_dict(0, 0.05), (80, 192, 192))
pipe = VolumePipeline(unet, icon, atlas)
base = [make_volume(i, shape) for i in range(4)]
vols = [Image(base[i % 4], **meta) for i in range(N)]
dev = torch.from_numpy(base[0]).cuda()
for _ in range(2): pipe.run(dev, vols[0], check=False)
torch.cuda.synchronize(); t = time.time()
for i in range(N): r = pipe.run(dev, vols[0], check=False)
torch.cuda.synchronize(); res_ms = (time.time() - t) / N * 1e3
print(f"resident loop: {res_ms:.2f} ms per volume")
for keep in (True, False):
    runner = CohortRunner(pipe, keep_on_device=keep)
    list(runner.run(vols[:3]))                       # warm-up (workspace allocation, first-use costs)
    torch.cuda.synchronize(); t = time.time(); st = []
    for _ in runner.run(vols): st.append(time.time() - t)   # results are consumed (not accumulated) as a cohort driver would
    torch.cuda.synchronize(); dt = time.time() - t
    steady = (st[N - 3] - st[3]) / (N - 6) * 1e3
    print(f"{N} volumes from host memory, results {'left on the device' if keep else 'copied back (5 tensors, 0.6 GB per volume)'}: "
          f"{dt:.3f} s -> {N / dt:.2f} volumes/s; steady state {steady:.2f} ms per volume = {res_ms / steady:.3f} of resident; stats {runner.stats}")
    runner.close()
