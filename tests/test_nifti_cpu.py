"""NIfTI-1 I/O: header layout against the published spec, ITK's LPS conventions, round trips."""
import gzip
import struct

import numpy as np
import pytest

from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.io_nifti import (NIFTI_INTENT_VECTOR, NiftiError, read_displacement_nifti, read_nifti,
                                         write_displacement_nifti, write_nifti)


def _rot(seed, flip=False):
    q, _ = np.linalg.qr(np.random.default_rng(seed).normal(size=(3, 3)))
    if (np.linalg.det(q) < 0) != flip:
        q[:, 2] *= -1
    return q


@pytest.mark.parametrize("ext", [".nii", ".nii.gz"])
@pytest.mark.parametrize("seed,flip", [(0, False), (1, True), (2, False)])
def test_round_trip_geometry(tmp_path, ext, seed, flip):
    rng = np.random.default_rng(seed)
    arr = rng.normal(size=(5, 7, 6)).astype(np.float32)
    img = Image(arr, spacing=[0.36, 0.37, 0.7], origin=rng.normal(size=3) * 50, direction=_rot(seed, flip))
    p = str(tmp_path / ("a" + ext))
    write_nifti(p, img)
    back = read_nifti(p)
    assert back.array.dtype == np.float32 and np.array_equal(back.array, arr)
    assert np.allclose(back.spacing, img.spacing, rtol=1e-6)
    assert np.allclose(back.origin, img.origin, rtol=1e-6, atol=1e-5)
    assert np.allclose(back.direction, img.direction, atol=1e-6)


def test_header_layout_and_ras_flip(tmp_path):
    """Byte offsets of the NIfTI-1.1 header; an LPS-identity ITK image has srow_x = (-sx,0,0,-ox) in RAS."""
    img = Image(np.arange(24, dtype=np.int16).reshape(2, 3, 4), spacing=[2, 3, 4], origin=[10, 20, 30])
    p = str(tmp_path / "h.nii")
    write_nifti(p, img)
    raw = open(p, "rb").read()
    assert struct.unpack("<i", raw[:4])[0] == 348 and raw[344:348] == b"n+1\0"
    assert struct.unpack("<8h", raw[40:56])[:4] == (3, 4, 3, 2)                 # dim: x fastest
    assert struct.unpack("<3h", raw[68:74])[1:] == (4, 16)                      # DT_INT16, bitpix
    assert struct.unpack("<f", raw[108:112])[0] == 352.0 and len(raw) == 352 + 48
    srow = np.array(struct.unpack("<12f", raw[280:328])).reshape(3, 4)
    assert np.allclose(srow, [[-2, 0, 0, -10], [0, -3, 0, -20], [0, 0, 4, 30]])
    assert np.array_equal(np.frombuffer(raw, "<i2", 24, 352), np.arange(24))


def _hand_built(endian, qform=True, slope=2.0, inter=1.0):
    hdr = bytearray(348)
    struct.pack_into(endian + "i", hdr, 0, 348)
    struct.pack_into(endian + "8h", hdr, 40, 3, 3, 2, 2, 1, 1, 1, 1)
    struct.pack_into(endian + "3h", hdr, 68, 0, 2, 8)                             # uint8
    struct.pack_into(endian + "8f", hdr, 76, -1.0, 0.5, 0.6, 0.7, 0, 0, 0, 0)     # qfac = -1
    struct.pack_into(endian + "3f", hdr, 108, 352.0, slope, inter)
    struct.pack_into(endian + "2h", hdr, 252, 1 if qform else 0, 0)
    struct.pack_into(endian + "6f", hdr, 256, 0.0, 0.0, 0.0, 1.0, 2.0, 3.0)       # identity quaternion, offset (1,2,3)
    hdr[344:348] = b"n+1\0"
    return bytes(hdr) + b"\0\0\0\0" + bytes(range(12))


@pytest.mark.parametrize("endian", ["<", ">"])
def test_reads_qform_scaling_and_both_byte_orders(tmp_path, endian):
    p = str(tmp_path / "q.nii.gz")
    with gzip.open(p, "wb") as f:
        f.write(_hand_built(endian))
    img = read_nifti(p)
    assert img.array.shape == (2, 2, 3)
    assert np.allclose(img.array.reshape(-1), np.arange(12) * 2.0 + 1.0)         # scl_slope / scl_inter
    assert np.allclose(img.spacing, [0.5, 0.6, 0.7])
    assert np.allclose(img.origin, [-1, -2, 3])                                  # RAS -> LPS
    assert np.allclose(img.direction, np.diag([-1, -1, -1]))                     # qfac = -1 flips z; LPS flips x, y


def test_no_transform_falls_back_to_pixdim(tmp_path):
    p = str(tmp_path / "n.nii")
    open(p, "wb").write(_hand_built("<", qform=False, slope=0.0))
    img = read_nifti(p, dtype=None)
    assert img.array.dtype == np.uint8 and np.allclose(img.spacing, [0.5, 0.6, 0.7]) and np.allclose(img.direction, np.diag([-1, -1, 1]))


def test_displacement_field_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    disp = rng.normal(size=(4, 5, 6, 3))
    ref = Image(np.zeros((4, 5, 6), np.float32), spacing=[1, 2, 3], origin=[5, 6, 7], direction=_rot(4))
    p = str(tmp_path / "d.nii.gz")
    write_displacement_nifti(p, disp, ref)
    raw = gzip.open(p, "rb").read()
    assert struct.unpack("<8h", raw[40:56])[:6] == (5, 6, 5, 4, 1, 3) and struct.unpack("<h", raw[68:70])[0] == NIFTI_INTENT_VECTOR
    back, grid = read_displacement_nifti(p)
    assert np.array_equal(back, disp) and np.allclose(grid.direction, ref.direction, atol=1e-6)
    # stored components are RAS: x and y negated, component axis slowest
    stored = np.frombuffer(raw, "<f8", disp.size, 352).reshape(3, 4, 5, 6)
    assert np.array_equal(stored[0], -disp[..., 0]) and np.array_equal(stored[2], disp[..., 2])


def test_bool_masks_and_errors(tmp_path):
    p = str(tmp_path / "m.nii.gz")
    mask = np.random.default_rng(0).random((3, 4, 5)) > 0.5
    write_nifti(p, Image(mask))
    assert np.array_equal(read_nifti(p, dtype=None).array, mask.astype(np.uint8))
    bad = str(tmp_path / "bad.nii")
    open(bad, "wb").write(b"\0" * 400)
    with pytest.raises(NiftiError):
        read_nifti(bad)
    with pytest.raises(NiftiError):
        write_nifti(bad, Image(np.zeros((2, 2), np.float32)))
    trunc = str(tmp_path / "t.nii")
    open(trunc, "wb").write(_hand_built("<")[:356])
    with pytest.raises(NiftiError):
        read_nifti(trunc)
