#!/usr/bin/env python
"""Generate tests/golden/segment_fullsize[_<case>].npz (``--case base|bn|dc|win``, see oai_analysis_2_amd.synth.FULLSIZE_CASES:
base = weight seed 0 / volume seed 42; bn = weight seed 1 with BN=True (networks.py:39) / volume 43; dc = weight seed 2 with every
conv bias + 1.0 (DC-heavy activations) / volume 44; win = weight seed 3 / an intensity-windowed volume with 5 % of the voxels at
exactly 0 and at exactly 1, dask_processing.py:10-26): the REFERENCE's own ``Segmenter3DInPatchClassWise.segment``
(oai_analysis/segmentation/segmenter.py:100-131) run once, on CPU, on a seeded 384x384x160 volume (BASELINE config 2 size).

Run here only (``python tests/golden/make_golden_fullsize.py``, ~6 minutes on 8 cores): it imports the reference from
/root/reference, which does not exist on the GPU box.  The fixture is DATA ONLY:

* ``fc_mask_bits`` / ``tc_mask_bits``  the reference's boolean maps (``if_output_prob_map=False``), np.packbits over the
  flattened (160,384,384) array  (2 x 2.9 MB before compression);
* ``fc_prob_s`` / ``tc_prob_s``        the reference's probability maps sampled at ``[z0::sz, y0::sy, x0::sx]``;
* ``near_idx`` / ``near_prob``        flat index (class*V + voxel) and reference probability of EVERY voxel with
  |p - 0.5| < 1e-4: a mask flip of an fp32-grade kernel can only happen there, and the test demands that a flipped voxel
  is in this list with |p_ref - 0.5| < 1e-5;
* ``prob_sum`` (f64 per class), ``mask_count``, ``volume_sha256`` (so the GPU box can prove it regenerated the same input).
"""
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = os.environ.get("OAI_REFERENCE", "/root/reference")

SHAPE = (160, 384, 384)
START, STRIDE = (1, 2, 3), (4, 4, 4)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="base", help="base | bn | dc | win (oai_analysis_2_amd.synth.FULLSIZE_CASES)")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    from make_golden import install_itk_shim
    install_itk_shim()
    sys.path.insert(0, REF)
    from oai_analysis.segmentation.segmenter import Segmenter3DInPatchClassWise  # the reference
    from oai_analysis_2_amd.synth import make_fullsize_case

    torch.set_num_threads(args.threads)
    sd, vol, case = make_fullsize_case(args.case, SHAPE)
    VOLUME_SEED, WEIGHT_SEED = case["volume_seed"], case["weight_seed"]
    patch, ovl = (128, 128, 32), (16, 16, 8)            # analysis_object.py:18-26 literals; patch_size of the model JSON
    res = {"volume_seed": np.int64(VOLUME_SEED), "weight_seed": np.int64(WEIGHT_SEED), "patch": np.asarray(patch),
           "overlap": np.asarray(ovl), "start": np.asarray(START), "stride": np.asarray(STRIDE),
           "volume_sha256": np.frombuffer(hashlib.sha256(vol.tobytes()).digest(), np.uint8)}
    t0 = time.time()
    with tempfile.TemporaryDirectory() as td:
        cfg = os.path.join(td, "cfg.pth.tar")
        with open(cfg, "w") as f:
            json.dump({"patch_size": list(patch), "model": "UNet",
                       "model_setting": {"in_channels": 1, "n_classes": 2, "bias": True, "BN": bool(case["bn"])}}, f)
        ck = os.path.join(td, "model.pth.tar")
        torch.save({"model_state_dict": sd, "epoch": 1, "best_score": 0.0}, ck)
        seg = Segmenter3DInPatchClassWise(mode="pred", config=dict(
            ckpoint_path=ck, training_config_file=cfg, device="cpu", batch_size=4,
            overlap_size=ovl, output_prob=True, output_itk=True))
        fc, tc = seg.segment(vol.copy(), if_output_prob_map=True, if_output_itk=True)
        fc, tc = np.asarray(fc), np.asarray(tc)
        assert fc.dtype == np.float64 and fc.shape == SHAPE
    print("reference segment(): %.0f s" % (time.time() - t0), flush=True)
    prob = np.stack([fc, tc])                                    # f64 holding f32 values (image_transforms.py:504)
    assert np.array_equal(prob, prob.astype(np.float32).astype(np.float64))
    # the boolean output of the reference is `sigmoid(x) > 0.5` on the same tiles (segmenter.py:121-124), i.e. prob > 0.5
    # inside the kept region and False in the zeroed frame: identical to thresholding the stitched map, which avoids a
    # second 6-minute pass.
    mask = prob > 0.5
    res["fc_mask_bits"], res["tc_mask_bits"] = np.packbits(mask[0].ravel()), np.packbits(mask[1].ravel())
    res["mask_count"] = mask.reshape(2, -1).sum(1).astype(np.int64)
    sl = tuple(slice(a, None, s) for a, s in zip(START, STRIDE))
    res["fc_prob_s"], res["tc_prob_s"] = fc[sl].astype(np.float32), tc[sl].astype(np.float32)
    res["prob_sum"] = prob.reshape(2, -1).sum(1)
    flat = prob.reshape(-1)
    near = np.flatnonzero(np.abs(flat - 0.5) < 1e-4)
    res["near_idx"], res["near_prob"] = near.astype(np.int64), flat[near].astype(np.float32)
    res["case"] = np.asarray(args.case)
    res["prob_range"] = np.asarray([prob[:, 8:-8, 16:-16, 16:-16].min(), prob.max()])
    np.savez_compressed(os.path.join(HERE, case["file"]), **res)
    print(case["file"], res["mask_count"], res["prob_sum"], res["prob_range"], len(near), res["fc_prob_s"].shape,
          os.path.getsize(os.path.join(HERE, case["file"])))


if __name__ == "__main__":
    main()
