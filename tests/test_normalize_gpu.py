"""GPU parity: device image_normalize (radix-select percentiles + window) vs numpy's np.percentile + the ITK functor."""
import numpy as np
import pytest
import torch

from oracle import normalize as onorm

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["uniform", "mri_like", "negatives", "ties", "tiny"])
def test_image_normalize_bit_exact(case):
    from oai_analysis_2_amd import ops
    rng = np.random.default_rng(5)
    if case == "uniform":
        a = (rng.random((40, 97, 101), dtype=np.float32) * 1000).astype(np.float32)
    elif case == "mri_like":       # a large background of exact zeros + a long intensity tail
        a = rng.gamma(2.0, 150.0, (64, 96, 96)).astype(np.float32)
        a[rng.random(a.shape) < 0.4] = 0.0
    elif case == "negatives":
        a = rng.normal(0.0, 50.0, (33, 65, 67)).astype(np.float32)
    elif case == "ties":
        a = rng.integers(0, 17, (32, 64, 64)).astype(np.float32)
    else:
        a = rng.random((3, 5, 7), dtype=np.float32)
    ref, (wmin, wmax) = onorm.image_normalize(a)
    got, win = ops.image_normalize(torch.from_numpy(a).cuda(), return_window=True)
    win = win.cpu().numpy()
    assert win[0] == wmin and win[1] == wmax                 # order statistics + numpy's float32 lerp: bit exact
    assert np.array_equal(got.cpu().numpy(), ref)            # windowed image bit exact
    assert got.min() >= 0.0 and got.max() <= 1.0
