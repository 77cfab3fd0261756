"""A/B of a result-preserving option on one box: full-size segmentation (160 tiles, fp16x3), alternating the values, with a bit-identity
check of the stitched maps (the raw blocks hold unspecified values where the frame is zeroed).   usage: python scripts/ab_option.py <option> <v0,v1,...> [reps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
eng = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
vol = torch.from_numpy(make_volume(0)).cuda()
ref = None
OPT = sys.argv[1]; VALS = [int(v) for v in sys.argv[2].split(",")]; REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for rep in range(REPS):
    for v in VALS:
        eng.set_option(OPT, v)
        eng.segment_tiles(vol, (32, 128, 128), (8, 16, 16), None, 0, 160, (8, 16, 16)); torch.cuda.synchronize()
        t = time.time()
        for _ in range(3):
            out = eng.segment_tiles(vol, (32, 128, 128), (8, 16, 16), None, 0, 160, (8, 16, 16))
        torch.cuda.synchronize()
        dt = (time.time() - t) / 3
        maps = eng.stitch(out, vol.shape, (32, 128, 128), (8, 16, 16), (8, 16, 16))
        if ref is None: ref = maps.clone()
        print(f"{OPT}={v}: {dt*1e3:.1f} ms per volume (segmentation only), equal to first: {torch.equal(maps, ref)}", flush=True)
