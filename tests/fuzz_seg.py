"""Randomised parity of the default fp16x3 segmentation path (tiles, overlaps, ragged volumes, widths, BN) vs the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oracle import seg as oseg          # (a test helper: run by hand / from tests, never by the product)

rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
worst = 0.0
for case in range(int(os.environ.get("CASES", "12"))):
    wd = int(rng.choice([1, 1, 2, 4]))                                             # (ec0 needs cout % 8 == 0: width / 4 is the narrowest; width / 1 = the reference's 32 channels, the only width whose ec0 is computed inside ec1's halo staging)
    bn = bool(rng.integers(0, 2))
    tile = tuple(int(8 * rng.integers(1, 5)) for _ in range(3))                  # z,y,x multiples of 8 up to 32
    tile = (tile[0], tile[1] + 8 * int(rng.integers(0, 3)), tile[2] + 8 * int(rng.integers(0, 5)))
    ovl = tuple(int(rng.integers(1, max(2, t // 4 + 1))) for t in tile)             # >= 1 (an overlap of 0 zeroes the whole map in the reference)
    shape = tuple(int(rng.integers(t - 2 * o + 1, 3 * (t - 2 * o) + 2 * o)) for t, o in zip(tile, ovl))
    sd = make_unet_state_dict(seed=case, width_div=wd, bn=bn)
    vol = make_volume(100 + case, shape)
    patch, ov = tile[::-1], ovl[::-1]                                            # reference order (x,y,z)
    try:
        fc_ref, tc_ref = oseg.segment(vol, sd, patch, ov, output_prob=True)
    except Exception as e:
        print(f"case {case}: oracle rejects tile {tile} ovl {ovl} shape {shape}: {type(e).__name__}"); continue
    eng = UNetEngine(sd, precision="fp16x3")
    crop = (ov[2], ov[0], ov[1])
    b = eng.segment_tiles(torch.from_numpy(vol).cuda(), tile, ovl, out_mode=0, batch=int(rng.integers(1, 9)), crop_zyx=crop)
    maps = eng.stitch(b, shape, tile, ovl, crop).cpu().numpy()
    err = max(np.abs(maps[0] - fc_ref).max(), np.abs(maps[1] - tc_ref).max())
    worst = max(worst, err)
    print(f"case {case}: width/{wd} bn={bn} tile {tile} ovl {ovl} volume {shape}: max|dp| {err:.2e} overflow={eng.range_overflow()}")
    assert err < 1e-5, "fp16x3 differs from the oracle"
print("worst", worst)
