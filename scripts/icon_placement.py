"""Where should the registration run?  It needs only the image.  Today it starts with the segmentation on a side stream (pipeline._run_overlapped).
Experiment: delay its start on the side stream by a spin kernel (torch.cuda._sleep) so that it lands under later layers (the write-bound up-conv dc3 and
dc2, which is not power-limited), and compare the step time with: serial, overlapped from the start, no registration at all."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.pipeline import CROP_ZYX, OVERLAP_ZYX, TILE_ZYX, VolumePipeline
from oai_analysis_2_amd.registration import IconEngine
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume
SHAPE = (160, 384, 384)
unet = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
icon = IconEngine(make_icon_state_dict(0, last_scale=0.1))
atlas = Image(make_volume(1000, SHAPE), [0.36, 0.36, 0.7], [0.0, 0.0, 0.0])
pipe = VolumePipeline(unet, icon, atlas)
vnp = make_volume(0, SHAPE); vol = torch.from_numpy(vnp).cuda(); meta = Image(vnp, [0.36, 0.36, 0.7], [2.0, -3.0, 1.0])
unet.calibrate_volume(vol, TILE_ZYX, OVERLAP_ZYX, CROP_ZYX)
# cycles of torch.cuda._sleep per ms
torch.cuda.synchronize(); t = time.time(); torch.cuda._sleep(200_000_000); torch.cuda.synchronize(); cyc_per_ms = 200_000_000 / ((time.time() - t) * 1e3)
print(f"_sleep: {cyc_per_ms:.0f} cycles per ms")
def timeit(fn, n=4):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
print(f"segmentation + resample, no registration: {timeit(lambda: pipe.segment(vol)):.1f} ms (segment only)")
pipe.overlap_registration = False
print(f"serial: {timeit(lambda: pipe.run(vol, meta, check=False)):.1f} ms")
pipe.overlap_registration = True
print(f"overlapped from the start (shipped): {timeit(lambda: pipe.run(vol, meta, check=False)):.1f} ms")
orig = pipe.register
for delay_ms in (20, 40, 60, 80, 95, 105, 115, 125):
    def delayed(v, d=delay_ms):
        torch.cuda._sleep(int(d * cyc_per_ms))
        return orig(v)
    pipe.register = delayed
    print(f"registration delayed by {delay_ms} ms on the side stream: {timeit(lambda: pipe.run(vol, meta, check=False)):.1f} ms")
