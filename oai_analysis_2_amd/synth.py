"""Seeded synthetic inputs (volumes, U-Net / ICON weights, displacement fields).

There are no network assets in this environment (the reference downloads its
knee, atlas and weights with pooch, oai_analysis/data.py:8-22), so parity tests and
``bench.py`` run on seeded synthetic data of the reference's shapes.
Everything is generated with the torch *CPU* generator so that the container that
makes the golden fixtures and the GPU box produce bit-identical inputs.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F

# (name, kind, cin, cout) in reference order, networks.py:43-66.  Weight shapes:
#   'c3' Conv3d            [cout, cin, 3,3,3]      'c1' Conv3d [cout, cin, 1,1,1]
#   't3' ConvTranspose3d   [cin, cout, 3,3,3]      'up' ConvTranspose3d [cin, cout, 2,2,2]
UNET_SPEC = [
    ("ec0", "c3", 1, 32), ("ec1", "c3", 32, 64), ("ec2", "c3", 64, 64), ("ec3", "c3", 64, 128),
    ("ec4", "c3", 128, 128), ("ec5", "c3", 128, 256), ("ec6", "c3", 256, 256), ("ec7", "c3", 256, 512),
    ("dc9", "up", 512, 512), ("dc8", "t3", 768, 256), ("dc7", "t3", 256, 256),
    ("dc6", "up", 256, 256), ("dc5", "t3", 384, 128), ("dc4", "t3", 128, 128),
    ("dc3", "up", 128, 128), ("dc2", "t3", 192, 64), ("dc1", "t3", 64, 64),
]


def make_unet_state_dict(seed: int = 0, n_classes: int = 2, in_channels: int = 1, bias: bool = True,
                         bn: bool = False, width_div: int = 1) -> "OrderedDict[str, torch.Tensor]":
    """State dict with the reference ``UNet`` key names.

    He-scaled normal weights keep activations O(1) through the 18 ReLU layers so that the
    logits straddle 0 and the thresholded masks are a meaningful parity target.
    ``width_div`` shrinks every hidden width (tests of the generic kernels only; the
    reference network is ``width_div=1``).
    """
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def ch(c):
        return max(1, c // width_div)

    for name, kind, cin, cout in UNET_SPEC:
        cin_e = in_channels if name == "ec0" else ch(cin)
        cout_e = ch(cout)
        k = {"c3": 3, "t3": 3, "up": 2}[kind]
        shape = (cout_e, cin_e, k, k, k) if kind == "c3" else (cin_e, cout_e, k, k, k)
        taps = 27 if k == 3 else 1        # a k2s2 up-conv has one tap per output voxel
        std = math.sqrt(2.0 / (cin_e * taps))
        sd[f"{name}.0.weight"] = torch.randn(shape, generator=g) * std
        if bias:
            sd[f"{name}.0.bias"] = (torch.rand(cout_e, generator=g) - 0.5) * 0.1
        if bn:
            sd[f"{name}.1.weight"] = 0.5 + torch.rand(cout_e, generator=g)
            sd[f"{name}.1.bias"] = (torch.rand(cout_e, generator=g) - 0.5) * 0.2
            sd[f"{name}.1.running_mean"] = (torch.rand(cout_e, generator=g) - 0.5) * 0.2
            sd[f"{name}.1.running_var"] = 0.5 + torch.rand(cout_e, generator=g)
            sd[f"{name}.1.num_batches_tracked"] = torch.tensor(1000, dtype=torch.long)
    c_last = ch(64)
    sd["dc0.weight"] = torch.randn((n_classes, c_last, 1, 1, 1), generator=g) * math.sqrt(1.0 / c_last)
    if bias:
        sd["dc0.bias"] = (torch.rand(n_classes, generator=g) - 0.5) * 0.1
    return sd


def make_volume(seed: int, shape_zyx: Sequence[int] = (160, 384, 384)) -> np.ndarray:
    """Knee-like volume in [0,1]: smooth low-frequency field + 0.05*N(0,1) noise, clamped."""
    g = torch.Generator().manual_seed(1234 + seed)
    coarse = [max(2, int(math.ceil(s / 8))) for s in shape_zyx]
    low = torch.rand((1, 1, *coarse), generator=g)
    smooth = F.interpolate(low, size=tuple(shape_zyx), mode="trilinear", align_corners=True)[0, 0]
    noise = torch.randn(tuple(shape_zyx), generator=g) * 0.05
    return (smooth + noise).clamp_(0.0, 1.0).numpy().astype(np.float32)


def make_volume_windowed(seed: int, shape_zyx: Sequence[int] = (160, 384, 384), frac: float = 0.05) -> np.ndarray:
    """``make_volume`` stretched so that ``frac`` of the voxels sit at exactly 0 and ``frac`` at exactly 1: the dynamic range of an
    intensity-windowed knee (dask_processing.py:10-26 clips at two percentiles and rescales to [0,1])."""
    v = make_volume(seed, shape_zyx).astype(np.float64)
    lo, hi = np.quantile(v, [frac, 1.0 - frac])
    return np.clip((v - lo) / (hi - lo), 0.0, 1.0).astype(np.float32)


# Full-size parity cases (tests/golden/make_golden_fullsize.py makes one reference golden per case; tests/test_fullsize_gpu.py
# and bench.py regenerate the same inputs): network variant x volume variant.
FULLSIZE_CASES = {
    "base": dict(weight_seed=0, bn=False, bias_shift=0.0, volume_seed=42, windowed=False, file="segment_fullsize.npz"),
    "bn": dict(weight_seed=1, bn=True, bias_shift=0.0, volume_seed=43, windowed=False, file="segment_fullsize_bn.npz"),
    "dc": dict(weight_seed=2, bn=False, bias_shift=1.0, volume_seed=44, windowed=False, file="segment_fullsize_dc.npz"),
    "win": dict(weight_seed=3, bn=False, bias_shift=0.0, volume_seed=45, windowed=True, file="segment_fullsize_win.npz"),
}


def make_fullsize_case(name: str, shape_zyx: Sequence[int] = (160, 384, 384)):
    """(state_dict, volume, case dict) of one full-size parity case.  ``bias_shift`` adds a constant to every conv bias of the
    17 ReLU layers (DC-heavy activations: every channel rides on a positive offset, the regime where the Winograd input
    transform's d1+d2 / d0-d2 terms cancel); ``bn`` uses the reference's ``BN=True`` network (networks.py:39)."""
    c = FULLSIZE_CASES[name]
    sd = make_unet_state_dict(seed=c["weight_seed"], bn=c["bn"])
    if c["bias_shift"]:
        for k in sd:
            if k.endswith(".0.bias"):
                sd[k] = sd[k] + c["bias_shift"]
    vol = (make_volume_windowed if c["windowed"] else make_volume)(c["volume_seed"], shape_zyx)
    return sd, vol, c


def make_smooth_field(seed: int, shape_zyx: Sequence[int], amplitude: float, coarse: int = 10) -> np.ndarray:
    """Smooth 3-channel displacement field [3,D,H,W] in [0,1] map units."""
    g = torch.Generator().manual_seed(4321 + seed)
    low = torch.randn((1, 3, coarse, coarse, coarse), generator=g)
    up = F.interpolate(low, size=tuple(shape_zyx), mode="trilinear", align_corners=True)[0]
    return (up * amplitude).numpy().astype(np.float32)


# ------------------------------------------------------------------------------------------------
# ICON (icon_registration.networks.tallUNet2) synthetic weights

ICON_DOWN = [2, 16, 32, 64, 256, 512]
ICON_UP_OUT = [16, 32, 64, 128, 256]
ICON_UP_IN = [48, 96, 192, 512, 512]      # down[1:] + (up_out[1:] + [0])


def make_icon_unet_state_dict(seed: int, dimension: int = 3, last_scale: float = 1.0, bn: str = "trained") -> "OrderedDict[str, torch.Tensor]":
    """Weights of one ``tallUNet2`` with the public package's parameter names.  ``bn``: "trained" = BatchNorm tensors as training leaves
    them (statistics moved, gamma / beta fitted, ``num_batches_tracked`` = 1000), "pristine" = as the constructor leaves them (a
    checkpoint of a package whose forward never calls the layers), "absent" = no BatchNorm keys at all.

    ``lastConv`` is zero-initialised in the real package and learnt; a trained net emits
    displacements of a few percent of the image extent, which ``last_scale`` reproduces.
    """
    g = torch.Generator().manual_seed(7000 + seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for d in range(5):
        cin, cout = ICON_DOWN[d], ICON_DOWN[d + 1]
        sd[f"downConvs.{d}.weight"] = torch.randn((cout, cin, 3, 3, 3), generator=g) * math.sqrt(1.0 / (cin * 27))
        sd[f"downConvs.{d}.bias"] = (torch.rand(cout, generator=g) - 0.5) * 0.1
        cin, cout = ICON_UP_IN[d], ICON_UP_OUT[d]
        # a k4 s2 p1 transposed conv touches 8 of its 64 taps per output voxel
        sd[f"upConvs.{d}.weight"] = torch.randn((cin, cout, 4, 4, 4), generator=g) * math.sqrt(1.0 / (cin * 8))
        sd[f"upConvs.{d}.bias"] = (torch.rand(cout, generator=g) - 0.5) * 0.1
        gamma, beta = 0.75 + 0.5 * torch.rand(cout, generator=g), (torch.rand(cout, generator=g) - 0.5) * 0.2      # (drawn in every mode:
        mean, var = (torch.rand(cout, generator=g) - 0.5) * 0.2, 0.75 + 0.5 * torch.rand(cout, generator=g)        # the conv weights do not depend on ``bn``)
        if bn == "pristine":
            gamma, beta, mean, var = torch.ones(cout), torch.zeros(cout), torch.zeros(cout), torch.ones(cout)
        elif bn not in ("trained", "absent"):
            raise ValueError(f"bn must be trained / pristine / absent, not {bn!r}")
        if bn != "absent":
            sd[f"batchNorms.{d}.weight"], sd[f"batchNorms.{d}.bias"] = gamma, beta
            sd[f"batchNorms.{d}.running_mean"], sd[f"batchNorms.{d}.running_var"] = mean, var
            sd[f"batchNorms.{d}.num_batches_tracked"] = torch.tensor(1000 if bn == "trained" else 0, dtype=torch.long)
    sd["lastConv.weight"] = torch.randn((3, 18, 3, 3, 3), generator=g) * (last_scale * math.sqrt(1.0 / (18 * 27)))
    sd["lastConv.bias"] = (torch.rand(3, generator=g) - 0.5) * 0.02 * last_scale
    return sd


# Step trees of the registration network as nested tuples: "u" = FunctionFromVectorField(tallUNet2), ("down", t) =
# DownsampleRegistration(t), ("two", phi, psi) = TwoStepRegistration(netPhi=phi, netPsi=psi).
ICON_TREES = {
    # SURVEY Appendix A's recollection of OAI_knees_gradICON_model: two half-resolution steps + one full-resolution step
    "3step": ("two", ("down", ("two", "u", "u")), "u"),
    # "the definition of our final 4 step registration network": one more full-resolution step around it (VERDICT r3 missing #2)
    "4step": ("two", ("two", ("down", ("two", "u", "u")), "u"), "u"),
    # the gradICON multi-resolution cascade: quarter-, half-, full-resolution
    "multires": ("two", ("down", ("two", ("down", "u"), "u")), "u"),
    # ... and with a last full-resolution step
    "multires4": ("two", ("two", ("down", ("two", ("down", "u"), "u")), "u"), "u"),
}


def icon_tree_prefixes(tree) -> list:
    """state_dict prefixes of the U-Nets of ``tree`` (a name of ICON_TREES or a nested tuple) in execution order, spelled as
    torch's ``nn.Module.state_dict`` spells the attribute path (``netPhi`` / ``netPsi`` of a TwoStep, ``net`` of a Downsample or FFVF)."""
    tree = ICON_TREES[tree] if isinstance(tree, str) and tree in ICON_TREES else tree
    out = []

    def visit(t, prefix):
        if t == "u":
            out.append(prefix + "net.")
        elif t[0] == "down":
            visit(t[1], prefix + "net.")
        elif t[0] == "two":
            visit(t[1], prefix + "netPhi.")
            visit(t[2], prefix + "netPsi.")
        else:
            raise ValueError(f"bad tree element {t!r}")

    visit(tree, "")
    return out


def make_icon_state_dict(seed: int = 0, last_scale: float = 0.3, tree="3step", bn: str = "trained") -> "OrderedDict[str, torch.Tensor]":
    """``regis_net`` state dict of a gradICON model: one tallUNet2 per FFVF of ``tree`` (default: the three-step tree, key prefixes
    ``netPhi.net.netPhi.net.*`` (u1, low-res), ``netPhi.net.netPsi.net.*`` (u2, low-res), ``netPsi.net.*`` (u3, full-res))."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for i, prefix in enumerate(icon_tree_prefixes(tree)):
        for k, v in make_icon_unet_state_dict(seed * 3 + i, last_scale=last_scale, bn=bn).items():
            out[prefix + k] = v
    return out


def identity_map(shape_dhw) -> np.ndarray:
    """[3,D,H,W] float32 map in ICON's [0,1] units: channel d = float32(index_d * (1 / (n_d - 1))) (the package's identity_map)."""
    D, H, W = (int(v) for v in shape_dhw)
    g = np.mgrid[0:D, 0:H, 0:W].astype(np.float64)
    for d, n in enumerate((D, H, W)):
        g[d] *= 1.0 / (n - 1)
    return g.astype(np.float32)


def write_standin_asset_tree(root: str, golden_npz: str) -> dict:
    """A stand-in for the reference's three release tarballs (oai_analysis/data.py:8-22, v2.0.0) in THEIR layout, from seeded synthetic data
    and one reference-made fixture -- so that the code path of BASELINE config 1 (``AnalysisObject()`` / ``Segmenter3DInPatchClassWise`` ->
    NIfTI in -> sum|d| against stored NIfTI maps, test/test_all.py:17-33) executes where the real assets cannot be downloaded:

        models/segmentation_model.pth.tar            torch.save({"model_state_dict": ...})  -- the seeded network of the fixture
        models/segmentation_train_config.pth.tar     the JSON training config (patch_size, model, model_setting)
        models/icon_weights.pth                      a seeded three-step gradICON regis_net state dict
        atlases/atlas_60_LEFT_baseline_NMI/atlas_image.nii.gz
        test_data/colab_case/image_preprocessed.nii.gz, FC_probmap.nii.gz, TC_probmap.nii.gz

    ``golden_npz`` = tests/golden/segment_small.npz: the REFERENCE's own ``Segmenter3DInPatchClassWise.segment`` output on the seeded
    24 x 72 x 72 volume (tests/golden/make_golden.py), which becomes the stored FC / TC maps.  It pins nothing new; it proves the day-one
    path runs.  Returns the config values a caller needs that the real tree fixes by convention (``overlap_size``: the fixture was made with
    (8, 8, 4) on 32 x 32 x 16 patches, the released network uses (16, 16, 8) on 128 x 128 x 32)."""
    import json
    import os
    from .image import Image
    from .io_nifti import write_nifti
    z = np.load(golden_npz)
    shape = tuple(int(v) for v in z["fc_prob"].shape)
    models = os.path.join(root, "models")
    atlas_dir = os.path.join(root, "atlases", "atlas_60_LEFT_baseline_NMI")
    case = os.path.join(root, "test_data", "colab_case")
    for d in (models, atlas_dir, case):
        os.makedirs(d, exist_ok=True)
    torch.save({"model_state_dict": make_unet_state_dict(seed=int(z["weight_seed"])), "epoch": 1}, os.path.join(models, "segmentation_model.pth.tar"))
    with open(os.path.join(models, "segmentation_train_config.pth.tar"), "w") as f:
        json.dump({"patch_size": [int(v) for v in z["patch"]], "model": "UNet",
                   "model_setting": {"in_channels": 1, "n_classes": 2, "bias": True, "BN": False}}, f)
    torch.save(make_icon_state_dict(0, last_scale=0.1), os.path.join(models, "icon_weights.pth"))
    spacing, origin = [0.36, 0.36, 0.7], [2.0, -3.0, 1.0]
    vol = make_volume(int(z["volume_seed"]), shape)
    write_nifti(os.path.join(case, "image_preprocessed.nii.gz"), Image(vol, spacing, origin))
    write_nifti(os.path.join(case, "FC_probmap.nii.gz"), Image(z["fc_prob"].astype(np.float32), spacing, origin))
    write_nifti(os.path.join(case, "TC_probmap.nii.gz"), Image(z["tc_prob"].astype(np.float32), spacing, origin))
    write_nifti(os.path.join(atlas_dir, "atlas_image.nii.gz"), Image(make_volume(1000, shape), spacing, [0.0, 0.0, 0.0]))
    info = {"overlap_size": [int(v) for v in z["overlap"]], "shape_zyx": list(shape), "source": os.path.basename(golden_npz),
            "note": "stand-in asset tree (oai_analysis_2_amd.synth.write_standin_asset_tree): seeded synthetic data + the reference's own output on it"}
    with open(os.path.join(root, "standin.json"), "w") as f:
        json.dump(info, f)
    return info
