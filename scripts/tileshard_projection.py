"""Tile-shard mode projected from ONE GPU: the wall time of every rank's share of a volume under the cost-balanced split of
parallel.tile_range_for_rank(160, r, N, costs) -- segment_tiles(range) and the z-slab of the phi-resample -- against 1/N of the whole-volume
pass, for N = 2, 4, 8.  The collectives themselves cannot be timed here; their sizes are printed (kept-centre blocks 189 MB all-gathered, one
94 MB broadcast, 2 x 94 MB of resampled slabs gathered).   python scripts/tileshard_projection.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.parallel import slab_range_for_rank, tile_range_for_rank
from oai_analysis_2_amd.pipeline import CROP_ZYX, OVERLAP_ZYX, TILE_ZYX, VolumePipeline
from oai_analysis_2_amd.registration import IconEngine
from oai_analysis_2_amd.segmentation.engine import UNetEngine
from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume

SHAPE = (160, 384, 384)
unet = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
for kv in os.environ.get("OPTIONS", "").split(","):
    if kv: unet.set_option(kv.split("=")[0], int(kv.split("=")[1]))
icon = IconEngine(make_icon_state_dict(0, 0.1))
atlas = Image(make_volume(1000, SHAPE), [0.36, 0.36, 0.7], [0.0, 0.0, 0.0])
pipe = VolumePipeline(unet, icon, atlas)
vol_np = make_volume(0, SHAPE)
vol = torch.from_numpy(vol_np).cuda()
meta = Image(vol_np, [0.36, 0.36, 0.7], [2.0, -3.0, 1.0])


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts), out


unet.calibrate_volume(vol, TILE_ZYX, OVERLAP_ZYX, CROP_ZYX)
t_full, blocks = timed(lambda: unet.segment_tiles(vol, TILE_ZYX, OVERLAP_ZYX, None, 0, None, CROP_ZYX))
t_stitch, maps = timed(lambda: unet.stitch(blocks, SHAPE, TILE_ZYX, OVERLAP_ZYX, CROP_ZYX))
t_reg, phi = timed(lambda: pipe.register(vol))
t_res, _ = timed(lambda: pipe.resample(maps, phi, meta))
t_vol, _ = timed(lambda: pipe.run(vol, meta, check=False), reps=3)
print(f"one GPU: segment 160 tiles {t_full:.2f} ms, stitch {t_stitch:.2f}, registration (side stream in run()) {t_reg:.2f}, both resamples {t_res:.3f}; "
      f"whole volume (pipe.run) {t_vol:.2f} ms")
costs = unet.tile_costs(SHAPE, TILE_ZYX, OVERLAP_ZYX, CROP_ZYX)
print("| N | rank | tiles | share of FLOPs | segment_tiles ms | x N / whole pass | resample slab ms |")
print("|---|---|---|---|---|---|---|")
for N in (2, 4, 8):
    worst, rows = 0.0, []
    for r in range(N):
        b, e = tile_range_for_rank(160, r, N, costs)
        t_r, _ = timed(lambda: unet.segment_tiles(vol, TILE_ZYX, OVERLAP_ZYX, (b, e), 0, None, CROP_ZYX))
        z0, z1 = slab_range_for_rank(SHAPE[0], r, N)
        t_s, _ = timed(lambda: pipe.resample(maps, phi, meta, (z0, z1)))
        worst = max(worst, t_r + t_s)
        rows.append((r, b, e, sum(costs[b:e]) / sum(costs), t_r, t_s))
    for r, b, e, sh, t_r, t_s in rows:
        print(f"| {N} | {r} | [{b}, {e}) = {e - b} | {sh:.4f} | {t_r:.2f} | {t_r * N / t_full:.3f} | {t_s:.3f} |")
    lat = worst + t_stitch
    print(f"| {N} | -> | | | slowest rank {worst:.2f} ms + stitch {t_stitch:.2f} = **{lat:.2f} ms** of compute per volume "
          f"(registration {t_reg:.2f} ms runs underneath on the side stream); one GPU: {t_vol:.2f} ms -> speed-up {t_vol / lat:.2f} of {N} "
          f"= efficiency {t_vol / lat / N:.2f} before the collectives (94 MB broadcast, 189 MB all-gather of blocks, 189 MB of slabs) | | |")
