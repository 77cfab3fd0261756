"""debug: the tap-pair form on ONE layer (LAYER=6: ec6), over level-3 sizes (round 5)"""
import os, sys, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine, LAYER_ORDER
from oracle import seg as oseg
def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
sd = make_unet_state_dict(seed=5, width_div=4)
layer = int(os.environ.get("LAYER", "6"))
for d3, h3, w3 in [(1, 2, 4), (1, 2, 3), (1, 2, 2), (1, 2, 5), (1, 2, 6), (1, 3, 4), (1, 1, 4), (2, 2, 4), (2, 2, 3), (3, 2, 4), (1, 2, 8), (1, 2, 7), (1, 4, 3), (2, 4, 2)]:
    shape = (8 * d3, 8 * h3, 8 * w3)
    x = torch.from_numpy(np.stack([make_volume(1, shape), make_volume(2, shape)]))[:, None]
    ref = oseg.unet_forward(x, sd).numpy()
    res = []
    for rep in range(2):
        eng = UNetEngine(sd, precision="fp16x3")
        eng.auto_calibrate = False
        eng.set_option("winograd", 0); eng.set_option("m16_layers", 1 << layer)
        res.append(rel(eng.forward_tiles(x.cuda()).cpu().numpy(), ref))
    print((d3, h3, w3), " ".join(f"{r:.1e}" for r in res), flush=True)
