"""CPU: pin the oracle (oracle/seg.py) against vectors produced by the reference itself
(tests/golden/make_golden.py imported /root/reference to make them)."""
import os

import numpy as np
import torch

from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume
from oracle import seg as oseg


def test_unet_small_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "unet_small.npz"))
    x = torch.from_numpy(z["x"])
    for bn in (False, True):
        sd = make_unet_state_dict(seed=int(z["seed"]), bn=bn)
        y = oseg.unet_forward(x, sd).numpy()
        ref = z["logits_bn%d" % int(bn)]
        # same ATen ops in the same order on the same machine: bit-identical here, and within
        # summation-order noise on another host CPU.
        np.testing.assert_allclose(y, ref, rtol=0, atol=2e-5 * np.abs(ref).max())


def test_partition_and_assemble_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "partition_cases.npz"))
    for idx in range(3):
        v = z[f"c{idx}_vol"]
        patch, ovl = tuple(z[f"c{idx}_patch"]), tuple(z[f"c{idx}_overlap"])
        tiles, g = oseg.partition(v, patch, ovl)
        assert np.array_equal(g["grid"], z[f"c{idx}_grid"])
        assert np.array_equal(tiles, z[f"c{idx}_tiles"])          # index work: bit exact
        asm = oseg.assemble(tiles[:, 0], g, crop_size_xyz=ovl)
        assert asm.dtype == np.float64
        assert np.array_equal(asm, z[f"c{idx}_assembled"])
        assert np.array_equal(oseg.assemble(tiles[:, 0], g, None), z[f"c{idx}_assembled_nocrop"])


def test_baseline_tile_geometry(golden_dir):
    z = np.load(os.path.join(golden_dir, "partition_cases.npz"))
    g = oseg.tile_geometry((160, 384, 384), (128, 128, 32), (16, 16, 8))
    assert np.array_equal(g["grid"], z["full_grid"]) and g["n_tiles"] == int(z["full_ntiles"]) == 160
    v = np.zeros((160, 384, 384), np.float32)
    v[::7, ::11, ::13] = 1.0
    tiles, _ = oseg.partition(v, (128, 128, 32), (16, 16, 8))
    assert np.array_equal(tiles.reshape(160, -1).sum(1), z["full_tile_sums"])


def test_segment_small_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "segment_small.npz"))
    vol = make_volume(int(z["volume_seed"]), (24, 72, 72))
    sd = make_unet_state_dict(seed=int(z["weight_seed"]))
    fc, tc = oseg.segment(vol, sd, tuple(z["patch"]), tuple(z["overlap"]), batch_size=4, output_prob=True)
    assert fc.dtype == np.float64 and fc.shape == (24, 72, 72)
    # the reference's own tolerance is sum|d| < 12 over 23.6M voxels (test/test_all.py:32-33)
    assert np.abs(fc - z["fc_prob"]).sum() < 12 * fc.size / 23592960 + 1e-3
    assert np.abs(tc - z["tc_prob"]).sum() < 12 * tc.size / 23592960 + 1e-3
    fm, tm = oseg.segment(vol, sd, tuple(z["patch"]), tuple(z["overlap"]), batch_size=4, output_prob=False)
    assert (fm != z["fc_mask"]).sum() <= 2 and (tm != z["tc_mask"]).sum() <= 2
    # frame of 4/8/8 voxels is exactly zero (image_transforms.py:509-513)
    assert fc[:4].max() == 0 and fc[:, :8].max() == 0 and fc[:, :, -8:].max() == 0


def test_trim_regions_and_flops():
    need = oseg.trim_regions()
    dims = {k: [h - l for l, h in zip(*v)] for k, v in need.items()}
    assert dims["dc1"] == [16, 96, 96] and dims["dc2"] == [18, 98, 98] and dims["dc3"] == [20, 100, 100]
    assert dims["dc4"] == [10, 50, 50] and dims["dc5"] == [12, 52, 52] and dims["dc6"] == [14, 54, 54]
    assert dims["dc7"] == [8, 28, 28] and dims["dc8"] == [8, 30, 30] and dims["dc9"] == [8, 32, 32]
    assert abs(oseg.unet_flops() / 1e9 - 976.94) < 0.05
    assert abs(oseg.unet_flops(trimmed=True) / 1e9 - 505.4) < 0.1


def test_fulltile_golden_matches_oracle(golden_dir):
    z = np.load(os.path.join(golden_dir, "unet_fulltile.npz"))
    vol = make_volume(int(z["volume_seed"]), (32, 128, 128))
    sd = make_unet_state_dict(seed=int(z["weight_seed"]))
    y = oseg.unet_forward(torch.from_numpy(vol)[None, None], sd)[0].numpy()
    ref = z["logits_centre"]
    np.testing.assert_allclose(y[:, 8:24, 16:112, 16:112], ref, rtol=0, atol=2e-5 * float(z["logits_abs_max"]))


def test_winograd_x_f32_emulation():
    """The numerics argument behind conv3_wino_f32 (the exact-fp32 path's default since round 6), as arithmetic on the CPU: one layer's sums emulated in float32
    in the kernels' orders (scripts/study/winograd_x_f32_error.py).  The x axis in Winograd F(2,3) form with a fresh partial sum per 8-channel chunk is no farther
    from float64 than the direct two-level form, and both are several times closer than one running sum (what a plain fp32 convolution computes)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("winograd_x_f32_error", os.path.join(root, "scripts", "study", "winograd_x_f32_error.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    chain, two, wino = mod.errors(64, X=2048)
    assert wino <= 1.05 * two and two < 0.6 * chain, (chain, two, wino)
