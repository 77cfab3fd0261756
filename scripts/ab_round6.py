"""Same-box A/B of round 6's bit-preserving kernel steps on the full-size segmentation pass (160 tiles, fp16x3): the ec0 shell kernel's pipelined walk
(option first_blocks: 4096 = one workgroup per 256 pairs, i.e. no walk and no prefetch, as rounds 1-5) and the up-conv's column-block walk (up_nbw: 1 =
one column block per workgroup, as rounds 1-5), alternated, with a bit-identity check of the stitched maps.   usage: python scripts/ab_round6.py [reps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oai_analysis_2_amd.segmentation.engine import UNetEngine
eng = UNetEngine(make_unet_state_dict(0), precision="fp16x3")
vol = torch.from_numpy(make_volume(0)).cuda()
T, O = (32, 128, 128), (8, 16, 16)
CONFIGS = [("round 5 forms (first_blocks 4096, up_nbw 1)", {"first_blocks": 4096, "up_nbw": 1}), ("round 6 defaults (first_blocks 24, up_nbw auto)", {"first_blocks": 24, "up_nbw": 0}),
           ("only the ec0 walk", {"first_blocks": 24, "up_nbw": 1}), ("only the up-conv walk", {"first_blocks": 4096, "up_nbw": 0})]
ref, acc = None, {n: [] for n, _ in CONFIGS}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for name, opts in CONFIGS:
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.segment_tiles(vol, T, O, None, 0, 160, O); torch.cuda.synchronize()
        t = time.time()
        for _ in range(3):
            out = eng.segment_tiles(vol, T, O, None, 0, 160, O)
        torch.cuda.synchronize()
        acc[name].append((time.time() - t) / 3 * 1e3)
        maps = eng.stitch(out, vol.shape, T, O, O)
        if ref is None: ref = maps.clone()
        assert torch.equal(maps, ref), name
for name, _ in CONFIGS:
    v = acc[name]
    print(f"{name:52s} {sum(v) / len(v):7.2f} ms per 160-tile pass  (runs: {' '.join(f'{x:.1f}' for x in v)})  maps bit-identical", flush=True)
