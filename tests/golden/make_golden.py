#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE ITSELF in this container.

Run here only (``python tests/golden/make_golden.py``): it imports the reference's Python
from /root/reference, which does not exist on the GPU box.  The reference's ``networks.py``
needs only torch; ``image_transforms.py`` / ``segmenter.py`` need three ``itk`` symbols
(GetArrayFromImage, GetImageFromArray, .CopyInformation), provided by the shim below
(SURVEY.md 8c).  Weights and volumes are the seeded synthetic ones of
``oai_analysis_2_amd.synth`` because the reference's assets are pooch downloads
(oai_analysis/data.py:8-22) and there is no network.

The fixtures are DATA ONLY: inputs (or their seeds) and the reference's outputs.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("OAI_REFERENCE", "/root/reference")


def install_itk_shim():
    class _Img(np.ndarray):
        def CopyInformation(self, other):
            return None

    itk = types.ModuleType("itk")
    itk.GetArrayFromImage = lambda im: np.asarray(im)
    itk.GetImageFromArray = lambda a: np.asarray(a).view(_Img)
    sys.modules["itk"] = itk


def make_vote_fixture(Partition):
    """Partition.assemble(is_vote=True) (image_transforms.py:466-484): 2- and 3-label tiles, with and without crop_size."""
    rng = np.random.default_rng(3)
    out = {}
    for idx, (shape, patch, ovl, nlab) in enumerate([((21, 40, 37), (16, 16, 8), (4, 4, 2), 2), ((9, 50, 33), (24, 20, 8), (2, 6, 1), 3)]):
        v = rng.random(shape, dtype=np.float32)
        p = Partition(patch, ovl, padding_mode="reflect", mode="pred")
        tiles = p({"image": v.copy(), "name": ""})["image"]
        lab = torch.from_numpy(rng.integers(0, nlab, size=tuple(tiles[:, 0].shape)))
        out[f"v{idx}_vol"], out[f"v{idx}_patch"], out[f"v{idx}_overlap"] = v, np.asarray(patch), np.asarray(ovl)
        out[f"v{idx}_labels"] = lab.numpy().astype(np.int8)
        out[f"v{idx}_vote"] = np.asarray(p.assemble(lab, is_vote=True, if_itk=False, crop_size=None))
        out[f"v{idx}_vote_crop"] = np.asarray(p.assemble(lab, is_vote=True, if_itk=False, crop_size=ovl))
        assert out[f"v{idx}_vote"].dtype == np.uint8 and out[f"v{idx}_vote_crop"].dtype == np.float64
    np.savez_compressed(os.path.join(HERE, "partition_vote.npz"), **out)
    print("partition_vote", {k: v.shape for k, v in out.items() if k.endswith("vote")})


def main():
    install_itk_shim()
    sys.path.insert(0, REF)
    from oai_analysis.segmentation.networks import UNet, get_network            # the reference
    from oai_analysis.segmentation.image_transforms import Partition           # the reference
    from oai_analysis.segmentation.segmenter import Segmenter3DInPatchClassWise  # the reference
    from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume

    torch.set_num_threads(8)
    torch.manual_seed(0)
    if "--only-vote" in sys.argv:
        return make_vote_fixture(Partition)

    # ---- 1. UNet forward, small tile, BN off / on -------------------------------------------
    out = {}
    g = torch.Generator().manual_seed(11)
    x = torch.rand((2, 1, 16, 32, 32), generator=g)
    out["x"] = x.numpy()
    for bn in (False, True):
        net = UNet(1, 2, bias=True, BN=bn)
        net.load_state_dict(make_unet_state_dict(seed=3, bn=bn), strict=True)
        net.eval()
        with torch.no_grad():
            y = net(x)
        out["logits_bn%d" % int(bn)] = y.numpy()
    out["seed"] = np.int64(3)
    np.savez_compressed(os.path.join(HERE, "unet_small.npz"), **out)
    print("unet_small", {k: getattr(v, "shape", v) for k, v in out.items()})

    # ---- 2. UNet forward, one full-size 32x128x128 tile (kept centre only) ------------------
    vol = make_volume(5, (32, 128, 128))
    net = UNet(1, 2, bias=True, BN=False)
    net.load_state_dict(make_unet_state_dict(seed=0), strict=True)
    net.eval()
    with torch.no_grad():
        y = net(torch.from_numpy(vol)[None, None])[0].numpy()
    np.savez_compressed(os.path.join(HERE, "unet_fulltile.npz"), volume_seed=np.int64(5), weight_seed=np.int64(0),
                        logits_centre=y[:, 8:24, 16:112, 16:112].astype(np.float32),
                        logits_abs_max=np.float32(np.abs(y).max()))
    print("unet_fulltile", y.shape, float(np.abs(y).max()))

    # ---- 3. Partition / assemble on ragged sizes ---------------------------------------------
    cases = {}
    rng = np.random.default_rng(7)
    for idx, (shape, patch, ovl) in enumerate([((21, 40, 37), (16, 16, 8), (4, 4, 2)),
                                               ((16, 32, 32), (32, 32, 16), (8, 8, 4)),
                                               ((9, 50, 33), (24, 20, 8), (2, 6, 1))]):
        v = rng.random(shape, dtype=np.float32)
        p = Partition(patch, ovl, padding_mode="reflect", mode="pred")
        tiles = p({"image": v.copy(), "name": ""})["image"]
        asm = np.asarray(p.assemble(tiles[:, 0], if_itk=False, crop_size=ovl))
        asm_nocrop = np.asarray(p.assemble(tiles[:, 0], if_itk=False, crop_size=None))
        cases[f"c{idx}_vol"] = v
        cases[f"c{idx}_patch"] = np.asarray(patch)
        cases[f"c{idx}_overlap"] = np.asarray(ovl)
        cases[f"c{idx}_tiles"] = tiles.numpy()
        cases[f"c{idx}_assembled"] = asm
        cases[f"c{idx}_assembled_nocrop"] = asm_nocrop
        cases[f"c{idx}_grid"] = np.asarray(p.tiles_grid_size)
    # the BASELINE geometry: numbers only
    p = Partition((128, 128, 32), (16, 16, 8), padding_mode="reflect", mode="pred")
    v = np.zeros((160, 384, 384), np.float32)
    v[::7, ::11, ::13] = 1.0
    tiles = p({"image": v, "name": ""})["image"]
    cases["full_grid"] = np.asarray(p.tiles_grid_size)
    cases["full_ntiles"] = np.int64(tiles.shape[0])
    cases["full_tile_sums"] = tiles.numpy().reshape(tiles.shape[0], -1).sum(1)
    np.savez_compressed(os.path.join(HERE, "partition_cases.npz"), **cases)
    print("partition_cases", cases["full_grid"], int(cases["full_ntiles"]))

    # ---- 4. the whole reference segment() on a small volume ----------------------------------
    vol = make_volume(9, (24, 72, 72))
    patch, ovl = (32, 32, 16), (8, 8, 4)
    res = {"volume_seed": np.int64(9), "weight_seed": np.int64(1), "patch": np.asarray(patch), "overlap": np.asarray(ovl)}
    with tempfile.TemporaryDirectory() as td:
        cfg = os.path.join(td, "cfg.pth.tar")   # the reference's config is JSON under a .pth.tar name
        with open(cfg, "w") as f:
            json.dump({"patch_size": list(patch), "model": "UNet",
                       "model_setting": {"in_channels": 1, "n_classes": 2, "bias": True, "BN": False}}, f)
        ck = os.path.join(td, "model.pth.tar")
        torch.save({"model_state_dict": make_unet_state_dict(seed=1), "epoch": 1, "best_score": 0.0}, ck)
        seg = Segmenter3DInPatchClassWise(mode="pred", config=dict(
            ckpoint_path=ck, training_config_file=cfg, device="cpu", batch_size=4,
            overlap_size=ovl, output_prob=True, output_itk=True))
        fc, tc = seg.segment(vol.copy(), if_output_prob_map=True, if_output_itk=True)
        res["fc_prob"], res["tc_prob"] = np.asarray(fc), np.asarray(tc)
        assert res["fc_prob"].dtype == np.float64
        fcm, tcm = seg.segment(vol.copy(), if_output_prob_map=False, if_output_itk=False)
        res["fc_mask"], res["tc_mask"] = np.asarray(fcm).astype(np.uint8), np.asarray(tcm).astype(np.uint8)
    res["fc_prob"] = res["fc_prob"].astype(np.float32)   # exact: the f64 maps hold f32 values
    res["tc_prob"] = res["tc_prob"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "segment_small.npz"), **res)
    print("segment_small", res["fc_prob"].shape, float(res["fc_prob"].max()), int(res["fc_mask"].sum()), int(res["tc_mask"].sum()))

    # ---- 4b. the label-vote branch of Partition.assemble ---------------------------------------
    make_vote_fixture(Partition)

    # ---- 5. reference registry quirk: unknown names return None (networks.py:858-862) ---------
    assert get_network("nope") is None and get_network("UNet") is UNet


if __name__ == "__main__":
    main()
