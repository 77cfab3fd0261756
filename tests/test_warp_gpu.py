"""GPU parity: HIP warp family (through the C ABI) vs the oracle's torch-CPU ops."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oai_analysis_2_amd.synth import make_smooth_field, make_volume
from oracle import icon as oicon
from oracle import resample as oresample

pytestmark = pytest.mark.gpu

TOL = 1e-4      # north_star: displacement fields within 1e-4 rel


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("shape,src_shape", [((24, 40, 36), (24, 40, 36)), ((24, 40, 36), (12, 20, 18)),
                                             ((17, 33, 29), (9, 17, 15)), ((160, 160, 160), (160, 160, 160))])
def test_grid_sample_and_compose(shape, src_shape):
    from oai_analysis_2_amd import ops
    img = make_volume(3, src_shape)
    field = make_smooth_field(4, src_shape, 0.03)
    coords = (oicon.identity_map(shape)[0].numpy() + make_smooth_field(5, shape, 0.05)).astype(np.float32)
    # push some coordinates outside [0,1] to exercise the border clamp
    coords[:, :2] -= 0.1
    coords[:, -2:] += 0.1
    ref_w = oicon.sample_at(torch.from_numpy(img)[None, None], torch.from_numpy(coords)[None])[0].numpy()
    ref_c = (torch.from_numpy(coords)[None] + oicon.sample_at(torch.from_numpy(field)[None], torch.from_numpy(coords)[None]))[0].numpy()
    got_w = ops.grid_sample3d(_dev(img[None]), _dev(coords)).cpu().numpy()
    got_c = ops.compose(_dev(field), _dev(coords)).cpu().numpy()
    assert _rel(got_w, ref_w) < TOL and np.abs(got_w - ref_w).max() < 2e-6
    assert _rel(got_c, ref_c) < TOL and np.abs(got_c - ref_c).max() < 2e-6


@pytest.mark.parametrize("shape,src_shape,amp", [((24, 40, 36), (24, 40, 36), 0.05), ((17, 33, 29), (9, 17, 15), 0.05), ((160, 160, 160), (160, 160, 160), 0.03),
                                                 ((40, 50, 70), (40, 50, 70), 0.4)])
def test_brick_form_is_bit_identical(shape, src_shape, amp):
    """Option "brick" (round 5): grid_sample3d / compose with the source box of a 16 x 8 x 4 output brick staged in LDS -- same corners, same
    weights, same order: torch.equal with the gather form, on smooth fields (boxes fit), on a field so rough that most boxes do NOT fit
    (amplitude 0.4: the per-brick fallback), with coordinates outside [0, 1] (border clamp) and with the identity map (coords None)."""
    from oai_analysis_2_amd import ops
    img = _dev(make_volume(3, src_shape)[None])
    field = _dev(make_smooth_field(4, src_shape, 0.03))
    coords = (oicon.identity_map(shape)[0].numpy() + make_smooth_field(5, shape, amp)).astype(np.float32)
    coords[:, :2] -= 0.1
    coords[:, -2:] += 0.1
    coords = _dev(coords)
    try:
        ops.warp_set_option("brick", 0)
        want = (ops.grid_sample3d(img, coords), ops.compose(field, coords), ops.grid_sample3d(img, None, out_shape=shape), ops.compose(field, None, out_shape=shape))
        ops.warp_set_option("brick", 1)
        got = (ops.grid_sample3d(img, coords), ops.compose(field, coords), ops.grid_sample3d(img, None, out_shape=shape), ops.compose(field, None, out_shape=shape))
    finally:
        ops.warp_set_option("brick", 0)
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_identity_paths():
    from oai_analysis_2_amd import ops
    shape, low = (20, 48, 44), (10, 24, 22)
    d_lo = make_smooth_field(7, low, 0.04)
    d_hi = make_smooth_field(8, shape, 0.04)
    ident = oicon.identity_map(shape)
    # sampled path at another resolution: id_h + sample(d_lo, id_h)
    ref = (ident + oicon.sample_at(torch.from_numpy(d_lo)[None], ident))[0].numpy()
    got = ops.compose(_dev(d_lo), None, out_shape=shape).cpu().numpy()
    assert np.abs(got - ref).max() < 2e-6
    # shortcut path: identity + d, bit exact
    ref2 = (ident + torch.from_numpy(d_hi)[None])[0].numpy()
    got2 = ops.compose(_dev(d_hi), None, shortcut=True).cpu().numpy()
    assert np.array_equal(got2, ref2)
    # warping with the identity map returns the image (D-5 of SURVEY Appendix D)
    img = make_volume(1, shape)
    got3 = ops.grid_sample3d(_dev(img[None]), None, out_shape=shape).cpu().numpy()[0]
    assert np.abs(got3 - img).max() < 1e-6


@pytest.mark.parametrize("shape", [(80, 192, 192), (5, 12, 11), (3, 6, 6)])
def test_avgpool_ceil(shape):
    from oai_analysis_2_amd import ops
    x = np.stack([make_volume(2, shape), make_volume(3, shape)])
    ref = F.avg_pool3d(torch.from_numpy(x)[None], 2, ceil_mode=True)[0].numpy()
    got = ops.avgpool2(_dev(x)).cpu().numpy()
    assert got.shape == ref.shape and np.abs(got - ref).max() < 1e-6


@pytest.mark.parametrize("src,dst", [((160, 384, 384), (80, 192, 192)), ((30, 50, 41), (40, 48, 48)), ((20, 24, 24), (40, 48, 48))])
def test_resize_trilinear(src, dst):
    from oai_analysis_2_amd import ops
    x = make_volume(6, src)
    ref = F.interpolate(torch.from_numpy(x)[None, None], size=dst, mode="trilinear", align_corners=False)[0, 0].numpy()
    got = ops.resize_trilinear(_dev(x[None]), dst).cpu().numpy()[0]
    assert np.abs(got - ref).max() < 2e-6


def test_phi_to_displacement_and_resample():
    from oai_analysis_2_amd import ops
    from oai_analysis_2_amd.image import Image
    from oai_analysis_2_amd.registration import resample_affines
    net = (20, 48, 44)
    phi = oicon.identity_map(net) + torch.from_numpy(make_smooth_field(2, net, 0.05))[None]
    ref_disp = oicon.displacement_itk(phi)
    got_disp = ops.phi_to_itk_displacement(phi[0].cuda())
    assert np.abs(got_disp.cpu().numpy() - ref_disp).max() < 1e-4
    A = Image(make_volume(1, (40, 90, 96)), [0.36, 0.37, 0.7], [1.0, 2.0, 3.0])
    th = 0.1
    rot = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    B = Image(np.zeros((36, 100, 88), np.float32), [0.4, 0.35, 0.75], [0.0, -1.0, 2.0], rot)
    ref = oresample.resample_through_phi(A.array.astype(np.float64), ref_disp, A, B)
    b2n, n2a = resample_affines(A, B, net)
    got = ops.resample_through_disp(_dev(A.array), got_disp, b2n, n2a, B.array.shape).cpu().numpy()
    assert (ref == 0).any() and (ref != 0).any()      # exercises the outside-buffer default pixel
    assert np.abs(got - ref).max() < 1e-5


@pytest.mark.parametrize("shape", [(20, 44, 36), (17, 33, 29), (80, 192, 192)])
def test_fused_warp_chain_is_bit_identical_to_the_op_by_op_closures(shape):
    """oai_warp_chain (SURVEY K15 "fuse chains", K18) against the sequence of compose / grid_sample3d launches it replaces in
    oai_icon_forward -- the three chains of regis_net_direction (oracle/icon.py) -- and against the oracle's torch ops."""
    from oai_analysis_2_amd import ops
    low = tuple((s + 1) // 2 for s in shape)
    d1, d2 = _dev(make_smooth_field(1, low, 0.03)), _dev(make_smooth_field(2, low, 0.02))
    d3 = _dev(make_smooth_field(3, shape, 0.02))
    A, a = _dev(make_volume(4, shape)), _dev(make_volume(5, low))
    # low-resolution warp through the isIdentity shortcut: a(id_l + d1)
    c_l = ops.compose(d1, None, shortcut=True)
    assert torch.equal(ops.warp_chain(low, start=d1, image=a), ops.grid_sample3d(a[None], c_l)[0])
    # A(c2), c2 = c1 + d1(c1), c1 = id_h + d2(id_h)
    c1 = ops.compose(d2, None, out_shape=shape, shortcut=False)
    c2 = ops.compose(d1, c1)
    got = ops.warp_chain(shape, fields=[d2, d1], image=A)
    assert torch.equal(got, ops.grid_sample3d(A[None], c2)[0])
    assert torch.equal(ops.warp_chain(shape, fields=[d2, d1]), c2) and torch.equal(ops.warp_chain(shape, fields=[d2]), c1)
    # phi = c4 + d1(c4), c4 = c3 + d2(c3), c3 = id_h + d3
    c3 = ops.compose(d3, None, shortcut=True)
    phi = ops.compose(d1, ops.compose(d2, c3))
    assert torch.equal(ops.warp_chain(shape, fields=[d2, d1], start=d3), phi)
    assert torch.equal(ops.warp_chain(shape, start=d3), c3)
    # and the oracle (torch CPU ops of the restated package)
    id_h = oicon.identity_map(shape)
    t = lambda x: x.cpu()[None]
    r3 = id_h + t(d3)
    r4 = r3 + oicon.sample_at(t(d2), r3)
    ref = (r4 + oicon.sample_at(t(d1), r4))[0].numpy()
    ident = id_h[0].numpy()
    assert _rel(phi.cpu().numpy() - ident, ref - ident) < TOL
    # longer chains (a four-step tree: start + three sampled fields of mixed resolution) against the op-by-op launches
    d4 = _dev(make_smooth_field(6, shape, 0.01))
    c = ops.compose(d4, None, shortcut=True)
    for f in (d3, d2, d1):
        c = ops.compose(f, c)
    assert torch.equal(ops.warp_chain(shape, fields=[d3, d2, d1], start=d4), c)
    assert torch.equal(ops.warp_chain(shape, fields=[d3, d2, d1], start=d4, image=A), ops.grid_sample3d(A[None], c)[0])
    with pytest.raises(Exception):
        ops.warp_chain(shape, fields=[d2, d1] * 5)
