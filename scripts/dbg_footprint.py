import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
tile, ovl, shape = (24, 40, 64), (6, 4, 8), (28, 66, 154)
sd = make_unet_state_dict(seed=50, width_div=1, bn=False)
v = torch.from_numpy(make_volume(200, shape)).cuda()
crop = (ovl[0], ovl[2], ovl[1])
eng = UNetEngine(sd, precision="fp16x3")
for kv in os.environ.get("OPTIONS", "").split(","):
    if kv: eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
eng.set_option("shared_enc", 0)
truth = eng.segment_tiles(v, tile, ovl, None, 2, 6, crop)
st = lambda blocks: eng.stitch(blocks, shape, tile, ovl, crop)
T = st(truth)
eng.set_option("shared_enc", 1)
for (b, e, batch) in ((0, 36, 6), (0, 36, 36), (0, 36, 1), (0, 36, 12), (0, 36, 4), (2, 36, 6), (5, 19, 6)):
    part = eng.segment_tiles(v, tile, ovl, (b, e), 2, batch, crop)
    bad = []
    for i in range(e - b):
        full = truth.clone(); full[b + i] = part[i]
        if not torch.equal(st(full), T): bad.append(b + i)
    print((b, e, batch), "tiles differing from the per-tile computation:", bad)
