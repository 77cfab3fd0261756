"""A/B of option "winograd" at full size: whole-volume segmentation time and the difference of the stitched maps (they are NOT bit-identical:
the x axis is computed in Winograd F(2,3) form -- same precision class, other rounding points).  WINO=1,3 picks the values compared with 0."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
eng = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
vol = torch.from_numpy(make_volume(0)).cuda()
tile, ovl = (32, 128, 128), (8, 16, 16)
def run(reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.time()
        b = eng.segment_tiles(vol, tile, ovl, None, 0, 160, ovl)
        torch.cuda.synchronize(); ts.append(time.time() - t)
    return min(ts), eng.stitch(b, vol.shape, tile, ovl, ovl)
eng.set_option("winograd", 0)                         # (the default is 3)
t0, base = run()
print(f"winograd 0: {t0 * 1e3:.1f} ms   flag {eng.range_flag()}")
for w in [int(v) for v in os.environ.get("WINO", "1,3").split(",")]:
    eng.set_option("winograd", w)
    t, got = run()
    d = (got - base).abs()
    flips = int(((got > 0.5) != (base > 0.5)).sum())
    print(f"winograd {w}: {t * 1e3:.1f} ms   max|dp| {d.max().item():.2e}  sum|dp| {d.double().sum().item():.3f}  mask flips {flips} of {base.numel()}  flag {eng.range_flag()}")
