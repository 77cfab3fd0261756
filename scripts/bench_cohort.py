"""BASELINE config 4: fused segment -> register -> resample per volume, a batch of 8 synthetic 384x384x160 volumes streamed
through one MI355X from HOST memory (upload of volume i+1 overlapped with the compute of volume i, results copied back):
the PCIe-inclusive rate that DESIGN.md 4 quotes beside the HBM-resident `value` of bench.py."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.cohort import CohortRunner
from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.pipeline import VolumePipeline
from oai_analysis_2_amd.registration import IconEngine
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume

N = int(os.environ.get("N", "8"))
shape = (160, 384, 384)
meta = dict(spacing=[0.36, 0.36, 0.7], origin=[0.0, 0.0, 0.0])
atlas = Image(make_volume(1000, shape), **meta)
unet = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
icon = IconEngine(make_icon_state_dict(0, 0.05), (80, 192, 192))
pipe = VolumePipeline(unet, icon, atlas)
vols = [Image(make_volume(i, shape), **meta) for i in range(N)]
for keep in (False, True):
    runner = CohortRunner(pipe, keep_on_device=keep)
    list(runner.run(vols[:2]))                       # warm-up (workspace allocation, first-use costs)
    torch.cuda.synchronize(); t = time.time()
    for _ in runner.run(vols): pass                  # results are consumed (not accumulated) as a cohort driver would
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"{N} volumes from host memory, results {'left on the device' if keep else 'copied back (5 tensors, 0.6 GB per volume)'}: "
          f"{dt:.3f} s -> {N / dt:.2f} volumes/s")
