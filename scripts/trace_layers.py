"""One full-size (160-tile) segmentation pass per repetition, for `rocprofv3 --kernel-trace`: the launch order gives the layer
(scripts/per_layer_table.py turns the trace into the per-layer table of profiles/<tag>_conv_per_layer.md).  PREC=fp16x3|f32."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
prec = os.environ.get("PREC", "fp16x3")
eng = UNetEngine(make_unet_state_dict(0), precision=prec)
vol = torch.from_numpy(make_volume(0)).cuda()
for kv in os.environ.get("OPTIONS", "").split(","):          # e.g. OPTIONS=wide=0,dead_stores=0
    if kv: eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for rep in range(int(os.environ.get("REPS", "3"))):
    torch.cuda.synchronize(); t = time.time()
    eng.segment_tiles(vol, (32, 128, 128), (8, 16, 16), None, 0, 160, (8, 16, 16))
    torch.cuda.synchronize(); print(prec, f"{(time.time() - t) * 1e3:.1f} ms")
