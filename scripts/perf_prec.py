"""Speed and error of the conv precision modes on the full-size golden tile + a 32-tile timing."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "unet_fulltile.npz"))
tile = torch.from_numpy(make_volume(int(z["volume_seed"]), (32, 128, 128)))[None, None].cuda()
vol = torch.from_numpy(make_volume(0)).cuda()
ref = z["logits_centre"]; amax = float(z["logits_abs_max"])
eng = UNetEngine(make_unet_state_dict(0))
for prec in ("f32", "bf16x6", "bf16x3", "fp16x3"):
    eng.set_precision(prec)
    got = eng.forward_tiles(tile).cpu().numpy()[0][:, 8:24, 16:112, 16:112]
    err = np.abs(got - ref).max() / amax
    p, pr = 1 / (1 + np.exp(-got.astype(np.float64))), 1 / (1 + np.exp(-ref.astype(np.float64)))
    flips = int(((got > 0) != (ref > 0)).sum())
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        eng.segment_tiles(vol, (32, 128, 128), (8, 16, 16), (0, 32), 0, 32)
        torch.cuda.synchronize(); dt = time.time() - t
    print(f"{prec:7s} logits rel err vs reference golden {err:.2e}  sum|dp| scaled to 23.6M voxels {np.abs(p - pr).mean() * 23.6e6:8.2f}  "
          f"sign flips {flips}/{ref.size}  32 tiles {dt*1e3:7.1f} ms -> {32/160/dt:.2f} vol/s (segmentation only)")
