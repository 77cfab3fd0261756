#!/usr/bin/env python
"""Generate tests/golden/segment_fullsize_truth.npz: for every full-size parity case (oai_analysis_2_amd.synth.FULLSIZE_CASES) the
REFERENCE's own ``UNet`` (oai_analysis/segmentation/networks.py:38-149, imported from /root/reference) run on six interior tiles of the
case's volume twice -- in float32, exactly as ``Segmenter3DInPatchClassWise.segment`` runs it, and in float64 (``model.double()``): the
same algorithm without fp32 rounding, "the truth".  Stored per case, at the kept-centre voxels that are also in the strided sample
of segment_fullsize[_case].npz (global z = 1 mod 4, y = 2 mod 4, x = 3 mod 4):

* ``<case>_tiles``   tile indices (the reference's z-major order)
* ``<case>_truth``   float64 [n_tiles, 2, 4, 24, 24]   sigmoid(fp64 logits)
* ``<case>_ref32``   float32 [n_tiles, 2, 4, 24, 24]   sigmoid(fp32 logits) -- asserted here to be BIT-IDENTICAL to the full reference
                     run stored in segment_fullsize[_case].npz at the same voxels
* ``<case>_ref_err`` float64 [2]  sum |ref32 - truth| per class scaled to 23 592 960 voxels: the reference's OWN fp32 rounding noise in the
                     unit of its acceptance test (test/test_all.py:32-33 accepts < 12 against maps stored from another machine)

Why: that budget is an absolute number in probability units, set for the released network.  A synthetic network with a wider logit
range carries proportionally more fp32 noise in every implementation (the reference included), so tests/test_fullsize_gpu.py scales
the budget of a case by ref_err(case) / ref_err(base) and additionally bounds the GPU path's distance from the truth by a small
multiple of the reference's own.  Run here only (~8 minutes on 8 cores); data only.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = os.environ.get("OAI_REFERENCE", "/root/reference")
SHAPE = (160, 384, 384)
TILES_IJK = [(1, 1, 1), (2, 2, 1), (4, 1, 2), (5, 2, 2), (7, 1, 1), (8, 2, 2)]       # interior tiles: nothing of their centre is in the zeroed frame


def main():
    from make_golden import install_itk_shim
    install_itk_shim()
    sys.path.insert(0, REF)
    from oai_analysis.segmentation.networks import UNet                     # the reference
    from oai_analysis_2_amd.synth import FULLSIZE_CASES, make_fullsize_case
    torch.set_num_threads(int(os.environ.get("THREADS", "8")))
    res = {}
    for case, c in FULLSIZE_CASES.items():
        t0 = time.time()
        sd, vol, _ = make_fullsize_case(case, SHAPE)
        full = np.load(os.path.join(HERE, c["file"]))
        # Partition's reflect padding for this geometry (image_transforms.py:409-415): lo = overlap, hi = eff * grid + 2 ovl - size - ovl
        padded = np.pad(vol, [(8, 8), (16, 16), (16, 16)], mode="reflect")
        model = UNet(in_channels=1, n_classes=2, bias=True, BN=bool(c["bn"]))
        model.load_state_dict(sd, strict=True)
        model.eval()
        model64 = UNet(in_channels=1, n_classes=2, bias=True, BN=bool(c["bn"])).double()
        model64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, strict=True)
        model64.eval()
        truth, ref32, idx = [], [], []
        for (i, j, k) in TILES_IJK:
            # the reference feeds its tiles four at a time (batch_size=4, segmenter.py:108-119) and ATen's CPU convolution rounds differently for
            # another batch size: run the same batch of four the full run used (tiles 4 m .. 4 m + 3), keep ours
            t_idx = 16 * i + 4 * j + k
            batch = []
            for t in range(4 * (t_idx // 4), 4 * (t_idx // 4) + 4):
                bi, bj, bk = t // 16, (t // 4) % 4, t % 4
                batch.append(np.ascontiguousarray(padded[16 * bi:16 * bi + 32, 96 * bj:96 * bj + 128, 96 * bk:96 * bk + 128]))
            x4 = torch.from_numpy(np.stack(batch))[:, None]
            x = x4[t_idx % 4:t_idx % 4 + 1]
            with torch.no_grad():
                p32 = torch.sigmoid(model(x4))[t_idx % 4, :, 8:24, 16:112, 16:112].numpy()
                p64 = torch.sigmoid(model64(x.double()))[0, :, 8:24, 16:112, 16:112].numpy()
            # kept-centre voxel (zz, yy, xx) sits at global (16 i + zz, 96 j + yy, 96 k + xx); the full-run sample is [1::4, 2::4, 3::4]
            sl = (slice(None), slice(1, None, 4), slice(2, None, 4), slice(3, None, 4))
            a32, a64 = p32[sl], p64[sl]
            gz, gy, gx = (16 * i + 1 - 1) // 4, (96 * j + 2 - 2) // 4, (96 * k + 3 - 3) // 4
            for cls, key in enumerate(("fc_prob_s", "tc_prob_s")):
                same = full[key][gz:gz + 4, gy:gy + 24, gx:gx + 24]
                assert np.array_equal(same, a32[cls]), f"{case} tile {(i, j, k)}: the tile run differs from the full reference run"
            truth.append(a64); ref32.append(a32.astype(np.float32)); idx.append(16 * i + 4 * j + k)
        truth, ref32 = np.stack(truth), np.stack(ref32)
        err = np.abs(ref32.astype(np.float64) - truth).sum(axis=(0, 2, 3, 4)) * (23592960.0 / (truth.shape[0] * truth[0, 0].size))
        res[f"{case}_tiles"], res[f"{case}_truth"], res[f"{case}_ref32"], res[f"{case}_ref_err"] = np.asarray(idx), truth, ref32, err
        print(f"{case}: reference fp32 vs its own fp64 run, sum|dp| per 23.6 M voxels = {err}, max {np.abs(ref32 - truth).max():.2e}  ({time.time() - t0:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "segment_fullsize_truth.npz"), **res)
    print(os.path.getsize(os.path.join(HERE, "segment_fullsize_truth.npz")))


if __name__ == "__main__":
    main()
