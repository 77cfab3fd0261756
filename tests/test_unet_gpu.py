"""GPU parity: the HIP U-Net (through the C ABI) vs the oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oracle import seg as oseg

pytestmark = pytest.mark.gpu

REL = 1e-4      # logits gate of SURVEY 8d: max-abs error relative to the tensor's max-abs


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("bn", [False, True])
def test_forward_tiles_matches_reference_golden(golden_dir, bn):
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    z = np.load(os.path.join(golden_dir, "unet_small.npz"))
    eng = UNetEngine(make_unet_state_dict(seed=int(z["seed"]), bn=bn))
    got = eng.forward_tiles(torch.from_numpy(z["x"]).cuda()).cpu().numpy()
    ref = z["logits_bn%d" % int(bn)]
    assert got.shape == ref.shape
    assert _rel(got, ref) < REL


@pytest.mark.parametrize("precision", ["f32", "bf16x6", "bf16x3", "fp16x3"])
@pytest.mark.parametrize("width_div,shape", [(4, (8, 16, 24)), (2, (16, 24, 40)), (4, (24, 40, 16))])
def test_forward_tiles_ragged_shapes_vs_oracle(width_div, shape, precision):
    """Channel counts that are not multiples of the K chunk (8 / 16) and spatial sizes that need every strip shape."""
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    sd = make_unet_state_dict(seed=5, width_div=width_div)
    x = torch.from_numpy(np.stack([make_volume(1, shape), make_volume(2, shape), make_volume(3, shape)]))[:, None]
    ref = oseg.unet_forward(x, sd).numpy()
    got = UNetEngine(sd, precision=precision).forward_tiles(x.cuda()).cpu().numpy()
    assert _rel(got, ref) < REL


@pytest.mark.parametrize("precision", ["f32", "bf16x6", "bf16x3", "fp16x3"])
def test_full_size_tile_matches_reference_golden(golden_dir, precision):
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    z = np.load(os.path.join(golden_dir, "unet_fulltile.npz"))
    vol = make_volume(int(z["volume_seed"]), (32, 128, 128))
    sd = make_unet_state_dict(seed=int(z["weight_seed"]))
    eng = UNetEngine(sd, precision=precision)
    got = eng.forward_tiles(torch.from_numpy(vol)[None, None].cuda()).cpu().numpy()[0]
    ref = z["logits_centre"]
    err = np.abs(got[:, 8:24, 16:112, 16:112] - ref).max() / float(z["logits_abs_max"])
    assert err < REL
    # the trimmed, gather-fused segment path on a one-tile volume vs the oracle on the same tile
    centre = np.ascontiguousarray(vol[8:24, 16:112, 16:112])
    blocks = eng.segment_tiles(torch.from_numpy(centre).cuda(), (32, 128, 128), (8, 16, 16), out_mode=2).cpu().numpy()
    tiles, g = oseg.partition(centre, (128, 128, 32), (16, 16, 8))
    assert tiles.shape[0] == 1
    ref2 = oseg.unet_forward(torch.from_numpy(tiles), sd).numpy()[:, :, 8:24, 16:112, 16:112]
    assert blocks.shape == ref2.shape
    assert _rel(blocks, ref2) < REL


@pytest.mark.parametrize("precision", ["f32", "bf16x6", "fp16x3"])
def test_segment_small_matches_reference_golden(golden_dir, precision):
    """All fp32-grade modes must meet the reference's own acceptance test (sum|dp| < 12 per 23.6M voxels)."""
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    z = np.load(os.path.join(golden_dir, "segment_small.npz"))
    vol = make_volume(int(z["volume_seed"]), (24, 72, 72))
    patch, ovl = tuple(int(v) for v in z["patch"]), tuple(int(v) for v in z["overlap"])
    tile_zyx, ovl_zyx = patch[::-1], ovl[::-1]
    eng = UNetEngine(make_unet_state_dict(seed=int(z["weight_seed"])), precision=precision)
    v = torch.from_numpy(vol).cuda()
    crop_zyx = (ovl[2], ovl[0], ovl[1])             # assemble indexes crop_size as (x,y,z): image_transforms.py:511-512
    blocks = eng.segment_tiles(v, tile_zyx, ovl_zyx, out_mode=0, batch=7)    # ragged last batch
    maps = eng.stitch(blocks, vol.shape, tile_zyx, ovl_zyx, crop_zyx).cpu().numpy().astype(np.float64)
    trimmed = eng.stitch(eng.segment_tiles(v, tile_zyx, ovl_zyx, out_mode=0, batch=7, crop_zyx=crop_zyx), vol.shape, tile_zyx, ovl_zyx, crop_zyx)
    assert np.array_equal(trimmed.cpu().numpy().astype(np.float64), maps)   # border-tile trimming is bit-identical
    # the reference's own acceptance test: sum|d| < 12 per 23.6M voxels (test/test_all.py:32-33)
    budget = 12.0 * vol.size / 23592960
    assert np.abs(maps[0] - z["fc_prob"]).sum() < budget
    assert np.abs(maps[1] - z["tc_prob"]).sum() < budget
    assert np.abs(maps[0] - z["fc_prob"]).max() < 1e-5
    # masks: sigmoid(x) > 0.5, identical except where the reference prob is within fp32 noise of 0.5
    mblocks = eng.segment_tiles(v, tile_zyx, ovl_zyx, out_mode=1, batch=16)
    masks = eng.stitch(mblocks, vol.shape, tile_zyx, ovl_zyx, crop_zyx).cpu().numpy()
    for got, ref, prob in ((masks[0], z["fc_mask"], z["fc_prob"]), (masks[1], z["tc_mask"], z["tc_prob"])):
        diff = got.astype(np.uint8) != ref
        assert diff.sum() <= 3, f"{diff.sum()} mask voxels differ"
        assert np.all(np.abs(prob[diff] - 0.5) < 1e-5)
    # tile-range sharding (the multi-GPU path) produces the same blocks
    # (block voxels beyond the image / inside the zeroed frame are unspecified, so compare what stitch makes of them)
    part = torch.cat([eng.segment_tiles(v, tile_zyx, ovl_zyx, (0, 30), 0, 8, crop_zyx), eng.segment_tiles(v, tile_zyx, ovl_zyx, (30, 75), 0, 8, crop_zyx)])
    assert np.array_equal(eng.stitch(part, vol.shape, tile_zyx, ovl_zyx, crop_zyx).cpu().numpy().astype(np.float64), maps)


def test_cohort_runner_matches_single_runs():
    from oai_analysis_2_amd.cohort import CohortRunner
    from oai_analysis_2_amd.image import Image
    from oai_analysis_2_amd.pipeline import VolumePipeline
    from oai_analysis_2_amd.registration import IconEngine
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    from oai_analysis_2_amd.synth import make_icon_state_dict
    shape, net = (24, 72, 72), (40, 48, 48)
    pipe = VolumePipeline(UNetEngine(make_unet_state_dict(1, width_div=2)), IconEngine(make_icon_state_dict(1, 0.1), net_shape=net),
                          Image(make_volume(10, shape), [0.4, 0.4, 0.8]), tile_zyx=(16, 32, 32), overlap_zyx=(4, 8, 8), crop_zyx=(4, 8, 8), batch=8)
    imgs = [Image(make_volume(20 + i, shape), [0.4, 0.4, 0.8], [i, 0, 0]) for i in range(3)]
    got = dict(CohortRunner(pipe).run(imgs))
    assert sorted(got) == [0, 1, 2]
    for i, img in enumerate(imgs):
        ref = pipe.run(torch.from_numpy(img.array).cuda(), img)
        assert torch.equal(got[i].fc, ref.fc.cpu()) and torch.equal(got[i].tc_atlas, ref.tc_atlas.cpu()) and torch.equal(got[i].phi, ref.phi.cpu())
    assert sorted(i for i, _ in CohortRunner(pipe).run(imgs, rank=1, world=2)) == [1]


def test_bf16x3_prob_maps_within_the_north_star_tolerance(golden_dir):
    """The 3-pass mode is faster but only ~2^-17 per product: logits/probabilities stay within 1e-4, masks differ only
    where the reference probability is within 1e-4 of 0.5; it does NOT promise the reference's sum|dp| < 12."""
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    z = np.load(os.path.join(golden_dir, "segment_small.npz"))
    vol = make_volume(int(z["volume_seed"]), (24, 72, 72))
    patch, ovl = tuple(int(v) for v in z["patch"]), tuple(int(v) for v in z["overlap"])
    tile_zyx, ovl_zyx, crop_zyx = patch[::-1], ovl[::-1], (ovl[2], ovl[0], ovl[1])
    eng = UNetEngine(make_unet_state_dict(seed=int(z["weight_seed"])), precision="bf16x3")
    v = torch.from_numpy(vol).cuda()
    maps = eng.stitch(eng.segment_tiles(v, tile_zyx, ovl_zyx, out_mode=0), vol.shape, tile_zyx, ovl_zyx, crop_zyx).cpu().numpy()
    assert np.abs(maps[0] - z["fc_prob"]).max() < 1e-4 and np.abs(maps[1] - z["tc_prob"]).max() < 1e-4
    masks = eng.stitch(eng.segment_tiles(v, tile_zyx, ovl_zyx, out_mode=1), vol.shape, tile_zyx, ovl_zyx, crop_zyx).cpu().numpy()
    for got, ref, prob in ((masks[0], z["fc_mask"], z["fc_prob"]), (masks[1], z["tc_mask"], z["tc_prob"])):
        diff = got.astype(np.uint8) != ref
        assert np.all(np.abs(prob[diff] - 0.5) < 1e-4)


@pytest.mark.parametrize("shape,patch,ovl", [((21, 45, 38), (32, 32, 16), (8, 8, 4)),      # ragged: hi padding != overlap
                                             ((9, 20, 70), (32, 16, 8), (4, 2, 1)),        # reflect pad wider than the tile centre
                                             ((16, 32, 32), (32, 32, 16), (8, 8, 4)),      # exactly one tile's centre... plus frame
                                             ((30, 72, 68), (32, 32, 16), (6, 5, 1)),      # x / y overlaps differ: crop_size[0] lands on y (:511-512)
                                             ((13, 57, 52), (32, 32, 8), (5, 1, 0))])      # an overlap of 0: numpy's [0:-0] is empty -> all-zero maps
@pytest.mark.parametrize("precision", ["f32", "fp16x3"])
def test_segment_ragged_volumes_vs_oracle(shape, patch, ovl, precision):
    """Partition edge cases of image_transforms.py:407-415 on the gather-fused path (no golden: oracle is pinned)."""
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    sd = make_unet_state_dict(seed=7, width_div=4)
    vol = make_volume(3, shape)
    fc_ref, tc_ref = oseg.segment(vol, sd, patch, ovl, output_prob=True)
    eng = UNetEngine(sd, precision=precision)
    tile_zyx, ovl_zyx, crop_zyx = patch[::-1], ovl[::-1], (ovl[2], ovl[0], ovl[1])
    blocks = eng.segment_tiles(torch.from_numpy(vol).cuda(), tile_zyx, ovl_zyx, out_mode=0, batch=5)
    maps = eng.stitch(blocks, shape, tile_zyx, ovl_zyx, crop_zyx).cpu().numpy()
    assert np.abs(maps[0] - fc_ref).max() < 1e-5 and np.abs(maps[1] - tc_ref).max() < 1e-5
    # border-tile trimming (the frame stitch zeroes is not computed) must not change a single stitched voxel
    blocks2 = eng.segment_tiles(torch.from_numpy(vol).cuda(), tile_zyx, ovl_zyx, out_mode=0, batch=5, crop_zyx=crop_zyx)
    maps2 = eng.stitch(blocks2, shape, tile_zyx, ovl_zyx, crop_zyx).cpu().numpy()
    assert np.array_equal(maps2, maps)
    if min(crop_zyx) > 0:
        assert eng.volume_flops(shape, tile_zyx, ovl_zyx, crop_zyx) < eng.volume_flops(shape, tile_zyx, ovl_zyx, None)
        assert maps[0][:crop_zyx[0]].max() == 0 and maps[0][:, :, -crop_zyx[2]:].max() == 0      # the zeroed frame
    else:
        assert fc_ref.max() == 0 and maps.max() == 0                                                # the reference's degenerate case, reproduced


def test_bad_arguments_raise():
    from oai_analysis_2_amd import _lib
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    eng = UNetEngine(make_unet_state_dict(seed=7, width_div=4))
    v = torch.zeros((16, 32, 32), device="cuda")
    with pytest.raises(_lib.OaiError, match="multiple of 8"):
        eng.segment_tiles(v, (12, 32, 32), (2, 8, 8))
    with pytest.raises(ValueError, match="overlap"):
        eng.segment_tiles(v, (16, 32, 32), (8, 8, 8))
    with pytest.raises(_lib.OaiError, match="tile range"):
        eng.segment_tiles(v, (16, 32, 32), (4, 8, 8), (0, 99))
    with pytest.raises(ValueError):
        eng.set_precision("fp8")
    sd = make_unet_state_dict(seed=7, width_div=4)
    sd.pop("dc4.0.weight")
    with pytest.raises(KeyError, match="dc4.0.weight"):                 # strict load, utils.py:29
        UNetEngine(sd)


def test_sharded_pipeline_equals_unsharded_on_one_rank():
    """world_size 1 through the same code path the multi-GPU tile shard uses (collective logic: tests/test_parallel_cpu.py)."""
    from oai_analysis_2_amd.image import Image
    from oai_analysis_2_amd.pipeline import VolumePipeline
    from oai_analysis_2_amd.registration import IconEngine
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    from oai_analysis_2_amd.synth import make_icon_state_dict
    shape, net = (24, 72, 72), (40, 48, 48)
    pipe = VolumePipeline(UNetEngine(make_unet_state_dict(1, width_div=2)), IconEngine(make_icon_state_dict(1, 0.1), net_shape=net),
                          Image(make_volume(10, shape)), tile_zyx=(16, 32, 32), overlap_zyx=(4, 8, 8), crop_zyx=(4, 8, 8), batch=8)
    vol = torch.from_numpy(make_volume(11, shape)).cuda()
    a, b = pipe.run(vol, Image(vol.cpu().numpy())), pipe.run_sharded(vol, Image(vol.cpu().numpy()))
    assert torch.equal(a.fc, b.fc) and torch.equal(a.tc_atlas, b.tc_atlas)


def test_fp16x3_reports_activations_outside_fp16_range():
    """Split-fp16 needs |activation| <= 65504; a violation is flagged (never silent) and the Segmenter reruns in fp32."""
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    sd = make_unet_state_dict(seed=7, width_div=4)
    x = torch.from_numpy(make_volume(1, (16, 32, 32)))[None, None].cuda()
    eng = UNetEngine(sd, precision="fp16x3")
    eng.forward_tiles(x)
    assert not eng.range_overflow()
    big = {k: (v * 1e6 if k == "ec0.0.weight" else v) for k, v in sd.items()}       # e0 ~ 1e6 > 65504
    eng2 = UNetEngine(big, precision="fp16x3")
    eng2.auto_calibrate = False                                                         # exponents all zero: round 2's behaviour
    eng2.forward_tiles(x)
    assert eng2.range_flag() & 1 and not eng2.range_overflow()                          # reported once, then reset
    eng2.set_precision("f32")
    ref = oseg.unet_forward(x.cpu(), big).numpy()
    assert _rel(eng2.forward_tiles(x).cpu().numpy(), ref) < REL                        # the exact mode is unaffected
    # calibrated (the default), the same checkpoint is simply inside the window: per-layer power-of-two activation exponents
    eng3 = UNetEngine(big, precision="fp16x3")
    got = eng3.forward_tiles(x).cpu().numpy()
    assert not eng3.range_overflow() and eng3.act_exponents()[0][0] < -5
    assert _rel(got, ref) < REL
    eng3.forward_tiles(x * 200.0)                                                       # ... until an input 200 x louder than the calibration input arrives
    assert eng3.range_flag() & 1


@pytest.mark.parametrize("opts", [{"sres_mrep": 2}, {"sres_ring": 1}, {"xcd_group": 0}, {"xcd_group": 7}, {"sres": 0}, {"fuse_first": 0}, {"b_lds": 1},
                                  {"wide": 0}, {"wide": 2}, {"dead_stores": 0}, {"census": 0}, {"shared_enc": 0}, {"winograd": 1}, {"winograd": 2}, {"winograd": 3}, {"winograd": 7}, {"winograd": 11}, {"winograd": 17}, {"winograd": 19}, {"winograd": 34}, {"winograd": 51}, {"m16": 0}, {"persistent": 1}, {"up_nbw": 3}, {"up_nbw": 64}, {"first_blocks": 1}, {"first_blocks": 4096}])
def test_split_fp16_kernel_variants_agree(golden_dir, opts):
    """The tuning variants of the default path (2 z slices per block, the six-slot plane ring, other XCD dealings, fp32-resident
    activations, ec0 as its own launch instead of inside ec1's halo staging, weight fragments through the workgroup's LDS ring, the
    8-wave double-buffered kernel of the Cout % 128 == 0 layers off / forced also for small launches, skip tensors written in full,
    no range census, ec0 -> ec1 per tile instead of once over the padded volume + a shell per tile, the k2s2 up-conv's workgroups walking three / all column blocks
    instead of one -- the small volume's automatic choice) accumulate in the same k order: identical stitched maps, and the golden tolerance of the default.
    "winograd" (the x axis of the plain layers in Winograd F(2,3) form; default 19 = both cout classes, the two-group form on 16x16x32 tap pairs; 3 = both on 32x32x16; bit 5 = the 64-cout layer on them too) against the direct form (0): same precision, other rounding points.
    "m16" 0 (round 5): the direct kernel of the layers with Cout % 128 != 0 on 32x32x16 taps instead of 16x16x32 tap pairs -- other rounding points too; the plane ring and the
    weight ring exist in the 32x32x16 form only and are compared there."""
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    z = np.load(os.path.join(golden_dir, "segment_small.npz"))
    vol = torch.from_numpy(make_volume(int(z["volume_seed"]), (24, 72, 72))).cuda()
    patch, ovl = tuple(int(v) for v in z["patch"]), tuple(int(v) for v in z["overlap"])
    tile_zyx, ovl_zyx, crop_zyx = patch[::-1], ovl[::-1], (ovl[2], ovl[0], ovl[1])
    sd = make_unet_state_dict(seed=int(z["weight_seed"]))

    def run(options=()):
        eng = UNetEngine(sd, precision="fp16x3")
        if "census" in dict(options):                             # the calibration needs the census: calibrate first, then switch the bookkeeping off
            eng.calibrate_volume(vol, tile_zyx, ovl_zyx, crop_zyx, batch=9)
        for k, v in dict(options).items():                        # explicit options on the handle: the library reads no environment
            eng.set_option(k, v)
        return eng.stitch(eng.segment_tiles(vol, tile_zyx, ovl_zyx, out_mode=0, batch=9, crop_zyx=crop_zyx), vol.shape, tile_zyx, ovl_zyx, crop_zyx).cpu().numpy()

    base_opts = {"wide": 2} if "wide" not in opts else {}       # (the 24 x 72 x 72 volume has too few workgroups for the default to pick the wide kernel)
    if any(k in opts for k in ("sres_mrep", "sres_ring", "b_lds", "winograd")):
        base_opts["winograd"] = 0                                   # these configurations run every layer through the direct kernels: compared with the direct form
    if any(k in opts for k in ("sres_ring", "b_lds")):
        base_opts["m16"] = 0                                        # ... in its 32x32x16 form (the rings have no tap-pair variant)
        opts = {"m16": 0, **opts}
    base = run(base_opts)
    got = run(opts)
    if "sres" in opts or "winograd" in opts or opts == {"m16": 0}:  # other activation format / x axis in Winograd form / other MFMA shape: same precision, other rounding points
        assert np.abs(got - base).max() < 1e-5 and not np.array_equal(got, base)
    else:
        assert np.array_equal(got, base)
    budget = 12.0 * vol.numel() / 23592960
    assert np.abs(got[0].astype(np.float64) - z["fc_prob"]).sum() < budget


@pytest.mark.parametrize("precision", ["f32", "fp16x3"])
def test_mask_is_the_fp32_sigmoid_predicate_not_the_sign_test(precision):
    """`sigmoid(x) > 0.5` and `x > 0` DIFFER for 0 < x <= ~8.94e-8 (fp32 sigmoid rounds to exactly 0.5).
    A head whose logits all fall within a few 1e-7 of zero populates that window densely; the mask output (out_mode 1, fused head on the default
    path) must equal torch's own `torch.sigmoid(logits) > 0.5` -- the reference's expression, segmenter.py:121-124 -- voxel for voxel."""
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    sd = make_unet_state_dict(seed=6, width_div=2)
    sd = dict(sd)
    sd["dc0.weight"] = sd["dc0.weight"] * 2e-7
    sd["dc0.bias"] = torch.tensor([4e-8, -4e-8])
    eng = UNetEngine(sd, precision=precision)
    shape, tile, ovl = (24, 72, 72), (16, 32, 32), (4, 8, 8)
    vol = torch.from_numpy(make_volume(3, shape)).cuda()
    out = {m: eng.stitch(eng.segment_tiles(vol, tile, ovl, out_mode=m, crop_zyx=ovl), shape, tile, ovl, ovl).cpu()
           for m in (0, 1, 2)}
    inner = (slice(None), slice(4, -4), slice(8, -8), slice(8, -8))
    logits, prob, mask = out[2][inner], out[0][inner], out[1][inner]
    window = ((logits > 0) & (logits <= 7.5e-8)).sum().item()       # safely inside (0, 8.94e-8]: 1 - x rounds to 1 - 2^-24 for any 1-ulp expf
    assert window > 1000, "the test must populate the D-4 window"
    ref_prob = torch.sigmoid(logits)                                  # torch CPU: the reference's own expression
    assert (prob - ref_prob).abs().max().item() <= 6e-8               # two correctly-rounded-to-1-ulp expf's may differ by one ulp of 0.5
    m = mask > 0.5
    assert torch.equal(m, prob > 0.5)                                  # the mask is the predicate on the kernel's own fp32 sigmoid
    # D-4 proper: everything up to 8.94e-8 rounds to exactly 0.5 -> NOT set, although the logit is positive; clearly positive
    # logits are set; in between (7.5e-8, 1.8e-7) one ulp of expf decides and both implementations are counted
    assert not m[logits <= 7.5e-8].any() and not (ref_prob[logits <= 7.5e-8] > 0.5).any()
    assert m[logits >= 1.8e-7].all()
    between = (logits > 7.5e-8) & (logits < 1.8e-7)
    differ = (m != (ref_prob > 0.5)).sum().item()
    disagree = ((logits > 0) != m).sum().item()
    print(f"[D-4 {precision}] {window} logits in (0, 7.5e-8]; sign test and sigmoid predicate disagree on {disagree} voxels; "
          f"{differ} of {int(between.sum())} voxels in (7.5e-8, 1.8e-7) differ from torch's sigmoid by the last ulp")
    assert disagree >= window                 # the sign test would be wrong on the whole window
    assert differ <= int(between.sum())
    assert (mask > 0.5).any() and not (mask > 0.5).all()


def test_persistent_workgroups_are_bit_identical():
    """Option "persistent" (round 5): the specialised 64-cout Winograd form (dc2) with ONE workgroup per CU pulling blocks from per-XCD counters, the
    staging waves one block ahead (conv3_wino_sres<..., PS>).  Which workgroup computes a block, and in which order, changes nothing about a block's own
    arithmetic: the maps are bit-identical -- on a ragged volume with strips, border tiles (trimmed boxes: tiles without blocks in a launch) and several
    batch sizes (few blocks per workgroup ... more workgroups than blocks)."""
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    tile, ovl, shape = (24, 40, 64), (6, 4, 8), (28, 66, 154)
    crop = (ovl[0], ovl[2], ovl[1])
    v = torch.from_numpy(make_volume(201, shape)).cuda()
    eng = UNetEngine(make_unet_state_dict(seed=51, width_div=1), precision="fp16x3")
    st = lambda b: eng.stitch(b, shape, tile, ovl, crop)
    ref = st(eng.segment_tiles(v, tile, ovl, None, 2, 6, crop))
    eng.set_option("persistent", 1)
    for batch in (1, 6, 36):
        assert torch.equal(st(eng.segment_tiles(v, tile, ovl, None, 2, batch, crop)), ref), batch
    eng.set_option("persistent", 0)
    assert torch.equal(st(eng.segment_tiles(v, tile, ovl, None, 2, 6, crop)), ref)


@pytest.mark.parametrize("wino", [19, 51])
def test_maps_do_not_depend_on_batching_or_tile_ranges(wino):
    """A voxel's arithmetic depends on the parity of its x only -- not on the batch its tile travels in, the launch box (the union of the
    batch's trimmed boxes), the strip that covers it or the tile range of the call.  Found by tests/fuzz_seg.py in round 4: a kernel form
    that existed for the main block shape only (option winograd bit 5, then) gave the strips another summation order, so four of 36 tiles
    changed bits with the batch size; since then bit 5 switches ALL launch shapes of the 64-cout layer (51 = the default 19 + bit 5).
    Geometry of that case: ragged volume, three z rows, tiles through the shared encoder pass."""
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    tile, ovl, shape = (24, 40, 64), (6, 4, 8), (28, 66, 154)
    crop = (ovl[0], ovl[2], ovl[1])
    v = torch.from_numpy(make_volume(200, shape)).cuda()
    eng = UNetEngine(make_unet_state_dict(seed=50, width_div=1), precision="fp16x3")
    eng.set_option("winograd", wino)
    st = lambda b: eng.stitch(b, shape, tile, ovl, crop)
    ref = st(eng.segment_tiles(v, tile, ovl, None, 2, 6, crop))
    for batch in (1, 4, 12, 36):
        assert torch.equal(st(eng.segment_tiles(v, tile, ovl, None, 2, batch, crop)), ref), batch
    for b, e, batch in ((0, 2, 6), (2, 36, 6), (5, 19, 7)):
        whole = eng.segment_tiles(v, tile, ovl, None, 2, 6, crop)
        whole[b:e] = eng.segment_tiles(v, tile, ovl, (b, e), 2, batch, crop)
        assert torch.equal(st(whole), ref), (b, e, batch)
    eng.set_option("shared_enc", 0)
    assert torch.equal(st(eng.segment_tiles(v, tile, ovl, None, 2, 5, crop)), ref)


def test_exact_fp32_winograd_form_against_the_direct_form_and_across_batches():
    """Round 6: the exact-fp32 path runs its k3 layers through conv3_wino_f32 (x axis in Winograd F(2,3) form, fp32 products, the direct kernel's
    two-level accumulation; option "winograd_f32", default 1).  Same precision, other rounding points than the direct form (0): the logits agree
    to fp32 noise; against float64 its error is the direct form's (tests/test_fullsize_gpu.py gates both against the reference).  And as on the
    fp16x3 path a voxel's bits do not depend on the batch, the launch box or the strip that covers it (ragged volume: main blocks, both strip
    shapes, border tiles with trimmed boxes, the fused MaxPool of ec1 / ec3 / ec5 in whole-tile launches)."""
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    tile, ovl, shape = (24, 40, 64), (6, 4, 8), (28, 66, 154)
    crop = (ovl[0], ovl[2], ovl[1])
    v = torch.from_numpy(make_volume(201, shape)).cuda()
    eng = UNetEngine(make_unet_state_dict(seed=51, width_div=1), precision="f32")
    st = lambda b: eng.stitch(b, shape, tile, ovl, crop)
    ref = st(eng.segment_tiles(v, tile, ovl, None, 2, 6, crop))
    for batch in (1, 5, 36):
        assert torch.equal(st(eng.segment_tiles(v, tile, ovl, None, 2, batch, crop)), ref), batch
    whole = eng.segment_tiles(v, tile, ovl, None, 2, 6, crop)
    whole[5:19] = eng.segment_tiles(v, tile, ovl, (5, 19), 2, 7, crop)
    assert torch.equal(st(whole), ref)
    eng.set_option("winograd_f32", 0)
    direct = st(eng.segment_tiles(v, tile, ovl, None, 2, 6, crop))
    scale = float(direct.abs().max())
    err = float((ref - direct).abs().max()) / scale
    assert 0.0 < err < 2e-5, err                 # two fp32 evaluations of a 17-layer network (logits): different bits, fp32 noise apart
    # whole-tile launches (no trimming: forward_tiles): the pooled levels come out of the fused epilogue on both paths
    x = torch.from_numpy(np.stack([make_volume(202, tile), make_volume(203, tile)]))[:, None].cuda()
    d = eng.forward_tiles(x)
    eng.set_option("winograd_f32", 1)
    w = eng.forward_tiles(x)
    assert 0.0 < float((w - d).abs().max()) / float(d.abs().max()) < 2e-5
