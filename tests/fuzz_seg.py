"""Randomised parity of the default fp16x3 segmentation path (tiles, overlaps, ragged volumes, widths, BN) vs the oracle."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oracle import seg as oseg          # (a test helper: run by hand / from tests, never by the product)

rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
PREC = os.environ.get("PRECISION", "fp16x3")     # PRECISION=f32: the exact-fp32 path (conv3_wino_f32) on the same random geometries, plus its independence of the batch size
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
    eng = UNetEngine(sd, precision=PREC)
    crop = (ov[2], ov[0], ov[1])
    b = eng.segment_tiles(torch.from_numpy(vol).cuda(), tile, ovl, out_mode=0, batch=int(rng.integers(1, 9)), crop_zyx=crop)
    maps_t = eng.stitch(b, shape, tile, ovl, crop)
    if PREC == "f32":
        again = eng.stitch(eng.segment_tiles(torch.from_numpy(vol).cuda(), tile, ovl, out_mode=0, batch=int(rng.integers(1, 9)), crop_zyx=crop), shape, tile, ovl, crop)
        assert torch.equal(again, maps_t), "f32 maps depend on the batch size"
    maps = maps_t.cpu().numpy()
    err = max(np.abs(maps[0] - fc_ref).max(), np.abs(maps[1] - tc_ref).max())
    worst = max(worst, err)
    print(f"case {case}: width/{wd} bn={bn} tile {tile} ovl {ovl} volume {shape}: max|dp| {err:.2e} overflow={eng.range_overflow()}")
    assert err < 1e-5, f"{PREC} differs from the oracle"
print("worst", worst)

# ---- geometries that take the shared encoder pass (ec0 -> ec1 once over the padded volume + a shell per tile): tile % (4, 8, 16) == 0,
# (tile - 2 overlap) % (4, 8, 16) == 0, overlap >= 4, reference width.  Random ragged volumes, batch sizes (partial z ranges of the pass) and tile
# ranges; the stitched maps must EQUAL the per-tile computation bit for bit and match the oracle.
worst = 0.0
for case in range(int(os.environ.get("SHARED_CASES", "10")) if PREC == "fp16x3" else 0):
    tile = (int(rng.choice([16, 24, 32])), int(rng.choice([24, 32, 40, 48])), int(rng.choice([32, 48, 64])))
    ovl = (int(rng.choice([4, 6])), int(rng.choice([4, 8])), 8)
    if any(t - 2 * o <= 0 or (t - 2 * o) % b for t, o, b in zip(tile, ovl, (4, 8, 16))):
        continue
    shape = tuple(int(rng.integers(t - 2 * o + 1, 3 * (t - 2 * o) + 2 * o)) for t, o in zip(tile, ovl))
    sd = make_unet_state_dict(seed=50 + case, width_div=1, bn=bool(rng.integers(0, 2)))
    vol = make_volume(200 + case, shape)
    v = torch.from_numpy(vol).cuda()
    crop = (ovl[0], ovl[2], ovl[1])                                               # the reference's crop order (as in the loop above)
    fc_ref, tc_ref = oseg.segment(vol, sd, tile[::-1], ovl[::-1], output_prob=True)
    eng = UNetEngine(sd, precision="fp16x3")
    batch = int(rng.integers(1, 12))
    res = {}
    for sh in (1, 0):
        eng.set_option("shared_enc", sh)
        res[sh] = eng.stitch(eng.segment_tiles(v, tile, ovl, out_mode=0, batch=batch, crop_zyx=crop), shape, tile, ovl, crop)
    ntiles = res[1].numel() and int(np.prod([-(-s // (t - 2 * o)) for s, t, o in zip(shape, tile, ovl)]))
    cut = int(rng.integers(1, max(2, ntiles)))
    eng.set_option("shared_enc", 1)
    parts = torch.cat([eng.segment_tiles(v, tile, ovl, (0, cut), 0, batch, crop), eng.segment_tiles(v, tile, ovl, (cut, ntiles), 0, batch, crop)]) if cut < ntiles else None
    eq_parts = parts is None or torch.equal(eng.stitch(parts, shape, tile, ovl, crop), res[1])
    maps = res[1].cpu().numpy()
    err = max(np.abs(maps[0] - fc_ref).max(), np.abs(maps[1] - tc_ref).max())
    worst = max(worst, err)
    print(f"shared case {case}: tile {tile} ovl {ovl} volume {shape} batch {batch} ({ntiles} tiles, split at {cut}): shared == per-tile {torch.equal(res[1], res[0])}, "
          f"tile ranges == whole {eq_parts}, max|dp| vs oracle {err:.2e}, flag {eng.range_flag()}")
    assert torch.equal(res[1], res[0]) and eq_parts and err < 1e-5
print("worst (shared)", worst)
