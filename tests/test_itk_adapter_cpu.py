"""The ITK adapters EXECUTE (VERDICT r5 missing #4): ``DisplacementTransform.to_itk()`` -- what ``ICON_Registration.register`` hands to the reference's
``itk.resample_image_filter(..., transform=phi_AB)`` (test/test_all.py:42-52, dask_processing.py:95-111; built like
``icon_registration.itk_wrapper.create_itk_transform``) -- and ``image.to_itk`` / ``as_image`` (image_transforms.py:515-517's CopyInformation carry), against
``tests/itk_standin.py``: a stand-in that implements ITK's documented setter / composition rules.  NOT a pin (ITK is not installed anywhere this runs); what it
proves is internal consistency: the composite transform the adapter builds maps every atlas point to the same patient point as the coordinate chain the HIP
resample kernel applies (``registration.resample_affines`` + trilinear displacement lookup), including outside the field's buffer."""
import sys

import numpy as np
import pytest

from tests import itk_standin


@pytest.fixture()
def itk(monkeypatch):
    mod = itk_standin.make_module()
    monkeypatch.setitem(sys.modules, "itk", mod)
    return mod


def _images():
    from oai_analysis_2_amd.image import Image
    rng = np.random.default_rng(3)
    th = 0.2
    rot = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]]) @ np.diag([1.0, -1.0, 1.0])    # oblique and flipped
    A = Image(rng.random((20, 48, 44), dtype=np.float32), [0.36, 0.41, 0.7], [3.0, -2.0, 1.0], rot)
    B = Image(rng.random((22, 40, 46), dtype=np.float32), [0.4, 0.37, 0.65], [-1.0, 4.0, 0.5], np.eye(3))
    return A, B


def test_displacement_transform_to_itk_is_the_map_the_resample_kernel_applies(itk):
    from oai_analysis_2_amd.registration import DisplacementTransform, resample_affines
    from oracle.resample import _trilinear_clamped
    A, B = _images()
    net = (10, 24, 24)
    rng = np.random.default_rng(5)
    disp = rng.normal(0.0, 1.5, size=(*net, 3))                                        # xyz components, network-voxel units (float64 like ITK's field)
    T = DisplacementTransform(disp, A, B)
    comp = T.to_itk()
    assert len(comp.queue) == 3
    (A1, b1), (A2, b2) = resample_affines(A, B, net)
    PB, oB = B.index_to_physical_affine()
    PA, oA = A.index_to_physical_affine()
    PA_inv = np.linalg.inv(PA)
    # atlas (B) indices: interior points, points whose network coordinate leaves the field's buffer (identity there), fractional points
    idx = np.concatenate([rng.uniform(0, 1, (200, 3)) * (B.size_xyz - 1), rng.uniform(-3, 3, (50, 3)), (B.size_xyz - 1) + rng.uniform(-2, 4, (50, 3))])
    worst = 0.0
    for i in idx:
        got_phys = comp.TransformPoint(PB @ i + oB)                                 # ITK: physical B -> physical A
        got_index = PA_inv @ (got_phys - oA)
        n = A1 @ i + b1                                                               # the kernel's chain: B index -> network index space
        shape_xyz = np.asarray(net[::-1], np.float64)
        if np.all(n >= -0.5) and np.all(n < shape_xyz - 0.5):
            n = n + _trilinear_clamped(disp, np.array([n[0]]), np.array([n[1]]), np.array([n[2]]))[0]
        want_index = A2 @ n + b2
        worst = max(worst, float(np.abs(got_index - want_index).max()))
    assert worst < 1e-9, worst


def test_centered_affine_setter_order_gives_the_network_affine(itk):
    """The one place where an ITK adapter silently goes wrong: MatrixOffsetTransformBase recomputes offset / translation in its setters.  The affine
    legs of to_itk() must come out as p_phys = M (x_net - c_net) + c_img whatever ITK does in between."""
    from oai_analysis_2_amd.registration import DisplacementTransform, network_affine
    A, B = _images()
    net = (10, 24, 24)
    T = DisplacementTransform(np.zeros((*net, 3)), A, B)
    comp = T.to_itk()
    aff_A, _, aff_B_inv = comp.queue
    for img, t, inverse in ((A, aff_A, False), (B, aff_B_inv, True)):
        M, c_net, c_img = network_affine(img, net)
        for x in np.random.default_rng(1).uniform(-5, 30, (20, 3)):
            want = M @ (x - c_net) + c_img
            if inverse:
                assert np.abs(t.TransformPoint(want) - x).max() < 1e-9
            else:
                assert np.abs(t.TransformPoint(x) - want).max() < 1e-9
    # zero displacement: the composite is affine(A) o affine(B)^-1
    p = np.array([1.0, 2.0, 3.0])
    assert np.abs(comp.TransformPoint(p) - aff_A.TransformPoint(aff_B_inv.TransformPoint(p))).max() < 1e-12


def test_image_round_trip_through_the_itk_adapters(itk):
    from oai_analysis_2_amd.image import Image, as_image, to_itk
    A, _ = _images()
    back = as_image(to_itk(A))
    assert isinstance(back, Image) and np.array_equal(back.array, A.array)
    assert np.array_equal(back.spacing, A.spacing) and np.array_equal(back.origin, A.origin) and np.array_equal(back.direction, A.direction)
    other = to_itk(Image(np.zeros((2, 3, 4), np.float32)))
    other.CopyInformation(to_itk(A))                                                   # image_transforms.py:516-517
    assert np.array_equal(as_image(other).direction, A.direction)
