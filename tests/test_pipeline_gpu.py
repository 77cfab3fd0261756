"""GPU: VolumePipeline / CohortRunner -- the fp16 range guard on every entry point, the fused two-map resample
(bit-identical to the separate displacement + per-map resample), the z-slab form of it, and BASELINE config 4 at size
(8 full-size volumes streamed through one GPU)."""
import numpy as np
import pytest
import torch

from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume

pytestmark = pytest.mark.gpu


def _small_pipe(unet_sd, precision="fp16x3"):
    from oai_analysis_2_amd.pipeline import VolumePipeline
    from oai_analysis_2_amd.registration import IconEngine
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    shape, net = (24, 72, 72), (40, 48, 48)
    atlas = Image(make_volume(10, shape), [0.4, 0.35, 0.75], [0.0, -1.0, 2.0])
    pipe = VolumePipeline(UNetEngine(unet_sd, precision=precision), IconEngine(make_icon_state_dict(1, last_scale=0.1), net_shape=net), atlas,
                          tile_zyx=(16, 32, 32), overlap_zyx=(4, 8, 8), crop_zyx=(4, 8, 8), batch=8)
    return pipe, shape


def test_fused_two_map_resample_is_bit_identical_to_the_separate_path():
    from oai_analysis_2_amd import ops
    from oai_analysis_2_amd.registration import resample_affines
    pipe, shape = _small_pipe(make_unet_state_dict(1, width_div=2))
    vol = make_volume(9, shape)
    meta = Image(vol, [0.36, 0.37, 0.7], [1.0, 2.0, 3.0])
    res = pipe.run(torch.from_numpy(vol).cuda(), meta)
    disp = ops.phi_to_itk_displacement(res.phi)
    b2n, n2a = resample_affines(meta, pipe.atlas, pipe.icon.net_shape)
    for got, src in ((res.fc_atlas, res.fc), (res.tc_atlas, res.tc)):
        sep = ops.resample_through_disp(src, disp, b2n, n2a, pipe.atlas.array.shape)
        assert torch.equal(got, sep)
    # z-slab shards tile the same result
    maps = torch.stack([res.fc, res.tc])
    nz = pipe.atlas.array.shape[0]
    parts = [pipe.resample(maps, res.phi, meta, (z0, z1)) for z0, z1 in ((0, 7), (7, 8), (8, nz))]
    assert torch.equal(torch.cat(parts, 1), torch.stack([res.fc_atlas, res.tc_atlas]))
    with pytest.raises(ValueError):
        pipe.resample(maps, res.phi, meta, (5, nz + 1))


def test_stitch_reads_a_ragged_gather_buffer_in_place_on_the_device():
    """``oai_stitch_blocks_ranged`` over a slot table with n_ranges > 1 had only run at world 1 (one range, nothing padded);
    the ragged, in-place layout an 8-rank all_gather leaves was checked over gloo on CPU tensors with a Python re-implementation of the slot
    lookup.  Here ONE GPU builds that layout from a real ``segment_tiles`` result -- ragged bounds, stride = the longest range, every slot's
    tail filled with NaN garbage, the ranges written straight into their slots (``segment_tiles(out=slot)``) -- and the stitch of the
    ``GatheredBlocks`` must equal the stitch of the compact tensor bit for bit."""
    from oai_analysis_2_amd import parallel
    from oai_analysis_2_amd.segmentation.engine import UNetEngine, tile_grid
    eng = UNetEngine(make_unet_state_dict(1, width_div=2), precision="fp16x3")
    shape, tile, ovl = (40, 100, 90), (16, 32, 32), (4, 8, 8)
    vol = torch.from_numpy(make_volume(3, shape)).cuda()
    eff, grid, n = tile_grid(shape, tile, ovl)
    compact = eng.segment_tiles(vol, tile, ovl, crop_zyx=ovl)
    want = eng.stitch(compact, shape, tile, ovl, ovl)
    assert n == 210
    # 8 ragged ranges (27 / 26 / ... like a cost-balanced split), one-tile ranges at both ends, an EMPTY range in the middle
    for bounds in ([0, 27, 53, 80, 106, 132, 158, 184, 210], [0, 1, 209, 210], [0, 105, 105, 210]):
        world = len(bounds) - 1
        stride = max(bounds[r + 1] - bounds[r] for r in range(world))
        buf = torch.full((world * stride, eng.n_classes, *eff), float("nan"), dtype=torch.float32, device=vol.device)
        g = parallel.GatheredBlocks(buf, bounds, stride)
        for r in range(world):
            if bounds[r + 1] > bounds[r]:
                got = eng.segment_tiles(vol, tile, ovl, (bounds[r], bounds[r + 1]), crop_zyx=ovl, out=g.slot(r))    # computed straight into the slot
                assert got.data_ptr() == g.slot(r).data_ptr()
        # (block voxels inside the frame that stitch zeroes are not computed: they keep the NaN fill here and hold allocator garbage in `compact` --
        #  the stitched maps are what must agree)
        assert torch.equal(eng.stitch(g, shape, tile, ovl, ovl), want), bounds
        for r in range(world):                                              # the tails of the slots were never written
            assert torch.isnan(buf[r * stride + bounds[r + 1] - bounds[r]: (r + 1) * stride]).all()
    assert not torch.isnan(want).any()


def test_pipeline_and_cohort_repeat_an_overflowing_volume_in_fp32():
    """Weights that push ec0 beyond 65504: fp16x3 must never return garbage -- run() repeats in fp32, run(check=False) hands
    the flag back, CohortRunner repeats at download time; results equal the fp32 engine's."""
    from oai_analysis_2_amd.cohort import CohortRunner
    sd = make_unet_state_dict(seed=7, width_div=2)
    big = {k: (v * 1e6 if k == "ec0.0.weight" else v) for k, v in sd.items()}
    pipe, shape = _small_pipe(big, "fp16x3")
    pipe.unet.auto_calibrate = False          # exponents all zero: a volume outside the window (calibrated, this checkpoint is simply fine)
    ref_pipe, _ = _small_pipe(big, "f32")
    vols = [make_volume(20 + i, shape) for i in range(3)]
    meta = Image(vols[0], [0.36, 0.37, 0.7], [1.0, 2.0, 3.0])
    v = torch.from_numpy(vols[0]).cuda()
    ref = ref_pipe.run(v, meta)
    assert ref.overflow is None and not ref.repeated_f32
    raw = pipe.run(v, meta, check=False)
    assert int(raw.overflow.item()) & 1
    res = pipe.run(v, meta)
    assert res.repeated_f32 and pipe.unet.precision == "fp16x3"
    assert torch.equal(res.fc, ref.fc) and torch.equal(res.tc_atlas, ref.tc_atlas)
    assert not pipe.unet.range_overflow()                                   # the snapshot cleared the flag
    for keep in (False, True):
        got = dict(CohortRunner(pipe, keep_on_device=keep).run([Image(a, meta.spacing, meta.origin) for a in vols]))
        assert sorted(got) == [0, 1, 2] and all(r.repeated_f32 for r in got.values())
        assert torch.equal(got[0].fc.cpu(), ref.fc.cpu()) and torch.isfinite(got[2].fc_atlas).all()
    # and a healthy network is not repeated
    ok_pipe, _ = _small_pipe(sd, "fp16x3")
    r = ok_pipe.run(v, meta)
    assert not r.repeated_f32 and int(r.overflow.item()) == 0


def test_cohort_of_8_full_size_volumes_streams_and_matches_single_runs():
    """BASELINE config 4: 8 synthetic 384x384x160 volumes streamed from host memory through one GPU (upload of i+1 and download of
    i-1 overlap the compute of i).  Every volume's results equal a plain pipe.run of that volume; frame / range properties hold."""
    from oai_analysis_2_amd.cohort import CohortRunner
    from oai_analysis_2_amd.pipeline import VolumePipeline
    from oai_analysis_2_amd.registration import IconEngine
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    shape = (160, 384, 384)
    atlas = Image(make_volume(1000, shape), [0.36, 0.36, 0.7], [0.0, 0.0, 0.0])
    pipe = VolumePipeline(UNetEngine(make_unet_state_dict(0), precision="fp16x3"), IconEngine(make_icon_state_dict(0, last_scale=0.1)), atlas)
    imgs = [Image(make_volume(i, shape), [0.36, 0.36, 0.7], [2.0, -3.0, 1.0]) for i in range(8)]
    seen = []
    for i, r in CohortRunner(pipe).run(imgs):
        seen.append(i)
        assert not r.repeated_f32
        for m in (r.fc, r.tc):
            m = m.numpy()
            assert m[:8].max() == 0 and m[-8:].max() == 0 and m[:, :16].max() == 0 and m[:, :, -16:].max() == 0
            assert 0 < m[8:-8, 16:-16, 16:-16].min() and m.max() < 1
        assert r.fc_atlas.shape == shape and torch.isfinite(r.fc_atlas).all() and r.phi.shape == (3, 80, 192, 192)
        if i in (0, 3, 7):                                                   # single-run equality on three of the eight
            one = pipe.run(torch.from_numpy(imgs[i].array).cuda(), imgs[i])
            for name in ("fc", "tc", "phi", "fc_atlas", "tc_atlas"):
                assert torch.equal(getattr(one, name).cpu(), getattr(r, name)), (i, name)
        if i == 3:
            # the tile-shard latency path on one rank (registration on the side stream underneath the sharded segmentation, joined
            # before the z-slab resample) gives the same five tensors as run() at the BASELINE size
            sh = pipe.run_sharded(torch.from_numpy(imgs[i].array).cuda(), imgs[i])
            for name in ("fc", "tc", "phi", "fc_atlas", "tc_atlas"):
                assert torch.equal(getattr(sh, name).cpu(), getattr(r, name)), ("sharded", name)
    assert seen == list(range(8))


def test_cohort_results_in_recycled_host_buffers_equal_fresh_ones():
    """``CohortRunner(result_pool=n)``: the same results, handed out in a ring of n pre-faulted host result sets (the download leg is 90 % page faults of
    fresh memory: profiles/r06_cohort.md); a result stays valid until n more have been yielded; n below LAG + 2 is refused."""
    from oai_analysis_2_amd.cohort import CohortRunner
    pipe, shape = _small_pipe(make_unet_state_dict(1, width_div=2))
    imgs = [Image(make_volume(20 + i, shape), [0.36, 0.37, 0.7], [1.0, 2.0, 3.0]) for i in range(9)]
    fresh = {i: r for i, r in CohortRunner(pipe).run(imgs)}
    with pytest.raises(ValueError):
        CohortRunner(pipe, result_pool=2)
    runner = CohortRunner(pipe, result_pool=4)
    seen = {}
    for i, r in runner.run(imgs):
        for name in ("fc", "tc", "phi", "fc_atlas", "tc_atlas"):
            assert torch.equal(getattr(r, name), getattr(fresh[i], name)), (i, name)
        seen[i] = r.fc.data_ptr()
    assert len(set(seen.values())) == 4                                   # four recycled sets, not nine allocations
    assert seen[0] == seen[4] == seen[8] and seen[1] == seen[5]


def test_registration_under_the_segmentation_equals_registration_alone():
    """The overlapped pipeline runs the ICON kernels on a side stream UNDERNEATH the MFMA convolution kernels.  Alternating two
    full-size volumes, phi of every overlapped run must be bit-identical to the registration run alone (nothing else on the GPU).
    Regression test of round 2's finding (profiles/r02_packed_fp32_hazard.md): with hipcc's SLP-packed fp32 arithmetic
    (v_pk_fma_f32 / v_pk_mul_f32) the fused warp-chain kernel returned wrong values in 16-lane groups in ~40 % of such runs --
    invisible when the same volume is repeated (the wrong lanes differ from run to run, the stale-data checks do not apply) and
    invisible to every single-kernel parity test.  The library is built without packed fp32 ops (build.py)."""
    from oai_analysis_2_amd.pipeline import VolumePipeline
    from oai_analysis_2_amd.registration import IconEngine
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    shape = (160, 384, 384)
    atlas = Image(make_volume(1000, shape), [0.36, 0.36, 0.7], [0.0, 0.0, 0.0])
    pipe = VolumePipeline(UNetEngine(make_unet_state_dict(0), precision="fp16x3"), IconEngine(make_icon_state_dict(0, last_scale=0.1)), atlas)
    vols = [torch.from_numpy(make_volume(i, shape)).cuda() for i in range(2)]
    meta = Image(make_volume(0, shape), [0.36, 0.36, 0.7], [2.0, -3.0, 1.0])
    clean = []
    for v in vols:
        clean.append(pipe.register(v).clone())
        torch.cuda.synchronize()
    seg = [pipe.segment(v).clone() for v in vols]
    assert pipe.overlap_registration
    bad = 0
    for trial in range(24):                                              # (enough alternations that an intermittent 1-in-12 fault cannot pass by chance)
        k = trial % 2
        r = pipe.run(vols[k], meta)
        torch.cuda.synchronize()
        bad += int(not torch.equal(r.phi, clean[k])) + int(not torch.equal(torch.stack([r.fc, r.tc]), seg[k]))
    res = [pipe.run(vols[t % 2], meta, check=False) for t in range(4)]          # and queued back to back, unsynchronised
    torch.cuda.synchronize()
    bad += sum(int(not torch.equal(r.phi, clean[t % 2])) for t, r in enumerate(res))
    assert bad == 0, f"{bad} overlapped runs differ from the clean ones"
