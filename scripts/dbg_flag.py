import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
SHAPE, TILE, OVL, CROP = (160, 384, 384), (32, 128, 128), (8, 16, 16), (8, 16, 16)
for w in (0, 3):
    eng = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
    eng.set_option("winograd", w)
    v = torch.from_numpy(make_volume(42, SHAPE)).cuda()
    for mode in (2, 0):
        eng.segment_tiles(v, TILE, OVL, out_mode=mode, crop_zyx=CROP)
        print("winograd", w, "mode", mode, "census", [f"{c:.0f}" for c in eng.census()], "flag", eng.range_flag(reset=True), "exp", eng.act_exponents()[0])
