"""Per-layer device time of the conv3 kernel for one precision (run under rocprofv3 --kernel-trace)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
prec = os.environ.get("PREC", "bf16x6")
eng = UNetEngine(make_unet_state_dict(0), precision=prec)
vol = torch.from_numpy(make_volume(0)).cuda()
for kv in os.environ.get("OPTIONS", "").split(","):          # e.g. OPTIONS=fuse_first=0
    if kv: eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
T = int(os.environ.get("TILES", "160"))                        # tiles per pass (160 = the whole volume, the launch size of bench.py)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    eng.segment_tiles(vol, (32, 128, 128), (8, 16, 16), (0, T), 0, T, (8, 16, 16))
    torch.cuda.synchronize(); print(prec, time.time() - t)
