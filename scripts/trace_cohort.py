"""8 volumes through CohortRunner (results copied back) for `rocprofv3 --kernel-trace --memory-copy-trace`: what sits in the gap between two volumes' kernels"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.cohort import CohortRunner
from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.pipeline import VolumePipeline
from oai_analysis_2_amd.registration import IconEngine
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume
shape = (160, 384, 384)
meta = dict(spacing=[0.36, 0.36, 0.7], origin=[0.0, 0.0, 0.0])
pipe = VolumePipeline(UNetEngine(make_unet_state_dict(0), precision="fp16x3"), IconEngine(make_icon_state_dict(0, 0.05), (80, 192, 192)), Image(make_volume(1000, shape), **meta))
base = [make_volume(i, shape) for i in range(2)]
vols = [Image(base[i % 2], **meta) for i in range(8)]
runner = CohortRunner(pipe)
list(runner.run(vols[:4]))
torch.cuda.synchronize()
print("MARK", time.time_ns())
for _ in runner.run(vols): pass
torch.cuda.synchronize()
runner.close()
