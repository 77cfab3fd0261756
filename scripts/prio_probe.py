"""Resident pipe.run loop: the volume's kernels on the default stream vs on a HIGH-priority stream with the ICON side stream left at the default
(= lowest) priority -- does the registration then stop costing the conv kernels ~2 ms of interference, and does it still finish in their tails?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.pipeline import VolumePipeline
from oai_analysis_2_amd.registration import IconEngine
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume
shape = (160, 384, 384)
meta = dict(spacing=[0.36, 0.36, 0.7], origin=[0.0, 0.0, 0.0])
pipe = VolumePipeline(UNetEngine(make_unet_state_dict(0), precision="fp16x3"), IconEngine(make_icon_state_dict(0, 0.05), (80, 192, 192)), Image(make_volume(1000, shape), **meta))
img = Image(make_volume(0, shape), **meta)
dev = torch.from_numpy(img.array).cuda()
N = 10
def loop():
    for _ in range(2): pipe.run(dev, img, check=False)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(N): pipe.run(dev, img, check=False)
    torch.cuda.synchronize(); return (time.time() - t) / N * 1e3
hi = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[1])
lo = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[0])
res = []
for rep in range(2):
    pipe._side = None
    res.append(("default stream, side = same priority", loop()))
    with torch.cuda.stream(hi):
        pipe._side = None
        res.append(("high-priority stream, side = high too", loop()))
        # force a LOW-priority side stream under the high-priority main stream
        class _S:  pass
        pipe._side = lo
        orig_prio = type(lo).priority
        old = pipe._run_overlapped
        def run_lo(vol, meta_A):
            main = torch.cuda.current_stream()
            lo.wait_stream(main)
            with torch.cuda.stream(lo):
                phi = pipe.register(vol); phi.record_stream(main)
            maps = pipe.segment(vol); flag = pipe._flag_snapshot()
            main.wait_stream(lo)
            am = pipe.resample(maps, phi, meta_A)
            from oai_analysis_2_amd.pipeline import VolumeResult
            return VolumeResult(maps[0], maps[1], phi, am[0], am[1], flag)
        pipe._run_overlapped = run_lo
        res.append(("high-priority stream, side = LOW priority", loop()))
        pipe._run_overlapped = old
    pipe.overlap_registration = False
    res.append(("default stream, registration serial", loop()))
    pipe.overlap_registration = True
for k, v in res: print(f"{k:45s} {v:.2f} ms per volume")
