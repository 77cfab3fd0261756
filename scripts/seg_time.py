"""Whole-volume segmentation time (min / median of REPS passes of 160 tiles, device-synchronised) for the library in OAI_LIB_PATH and the
options in OPTIONS=name=value,... -- the A/B building block (run the candidates alternately on ONE box: boxes differ by 1-2 %)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
eng = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
vol = torch.from_numpy(make_volume(0)).cuda()
tile, ovl = (32, 128, 128), (8, 16, 16)
eng.segment_tiles(vol, tile, ovl, None, 0, 160, ovl)            # calibration + warm-up under the default options
for kv in os.environ.get("OPTIONS", "").split(","):
    if kv: eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
ts = []
for _ in range(int(os.environ.get("REPS", "7")) + 1):
    torch.cuda.synchronize(); t = time.time()
    b = eng.segment_tiles(vol, tile, ovl, None, 0, 160, ovl)
    torch.cuda.synchronize(); ts.append(time.time() - t)
ts = sorted(ts[1:])
ck = float(eng.stitch(b, vol.shape, tile, ovl, ovl).double().sum())
print(f"{os.environ.get('TAG', '')} OPTIONS={os.environ.get('OPTIONS', '')}: min {ts[0] * 1e3:.2f} ms  median {ts[len(ts) // 2] * 1e3:.2f} ms  flag {eng.range_flag()}  checksum {ck:.6f}")
