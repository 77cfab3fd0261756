"""How long would ec0 -> ec1 take ONCE over the reflect-padded volume (176 x 416 x 416 as one tile)?  Run under rocprofv3 --kernel-trace:
the FIRST instantiation's duration in the second pass is the answer (the rest of that giant-tile network is irrelevant)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
eng = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
eng.auto_calibrate = False
vol = torch.from_numpy(make_volume(0)).cuda()
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    eng.segment_tiles(vol, (176, 416, 416), (8, 16, 16), None, 0, 1)
    torch.cuda.synchronize(); print(f"{(time.time() - t) * 1e3:.1f} ms", flush=True)
