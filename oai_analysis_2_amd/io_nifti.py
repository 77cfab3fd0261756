"""NIfTI-1 read / write for the formats either side of the hot path.

The reference reads its inputs with ``itk.imread(path, itk.F)`` (test/test_all.py:18-21: ``image_preprocessed.nii.gz``,
``FC_probmap.nii.gz``, ``TC_probmap.nii.gz``) and the Dask pipeline writes probability maps / displacement fields with
``itk.imwrite`` (dask_processing.py).  ITK is not part of this image, so this module implements the published NIfTI-1.1
layout directly (348-byte header, ``n+1`` single-file variant, optional gzip) with ITK's conventions at the boundary:

* ITK images live in LPS physical space, NIfTI affines in RAS: x and y of origin / direction flip sign (itkNiftiImageIO).
* ``Image.array`` is [z, y, x]; NIfTI stores x fastest -- the same memory order, no transpose.
* ``sform`` wins when ``sform_code > 0``, else ``qform`` (quaternion), else ``pixdim`` alone (identity direction).
* ``scl_slope`` / ``scl_inter`` are applied on read when slope != 0 (and not the identity).
* Vector images (displacement fields): dim[0] = 5, dim[5] = components, ``intent_code`` = 1007 (NIFTI_INTENT_VECTOR),
  which is how ITK stores ``itk.Image[itk.Vector[itk.D, 3], 3]``; components are the slowest-varying axis in the file.

Host-side I/O only; nothing here runs on the timed path.
"""
from __future__ import annotations

import gzip
import struct
from typing import Optional, Tuple

import numpy as np

from .image import Image

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).name: k for k, v in _DTYPES.items()}
NIFTI_INTENT_VECTOR = 1007
_LPS = np.diag([-1.0, -1.0, 1.0])


class NiftiError(ValueError):
    pass


def _open(path: str, mode: str):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def _quaternion_to_matrix(b: float, c: float, d: float, qfac: float) -> np.ndarray:
    a2 = 1.0 - (b * b + c * c + d * d)
    if a2 < 1e-7:                                   # 180-degree rotation: renormalise (nifti1_io.c nifti_quatern_to_mat44)
        n = 1.0 / np.sqrt(b * b + c * c + d * d)
        b, c, d, a = b * n, c * n, d * n, 0.0
    else:
        a = np.sqrt(a2)
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - c * c - b * b]])
    R[:, 2] *= -1.0 if qfac < 0 else 1.0
    return R


def _matrix_to_quaternion(R: np.ndarray) -> Tuple[float, float, float, float]:
    """(b, c, d, qfac) of a proper/improper rotation matrix (nifti_mat44_to_quatern, orthonormal input)."""
    R = np.array(R, dtype=np.float64)
    qfac = 1.0
    if np.linalg.det(R) < 0:
        R[:, 2] *= -1.0
        qfac = -1.0
    a = R[0, 0] + R[1, 1] + R[2, 2] + 1.0
    if a > 0.5:
        a = 0.5 * np.sqrt(a)
        b, c, d = 0.25 * (R[2, 1] - R[1, 2]) / a, 0.25 * (R[0, 2] - R[2, 0]) / a, 0.25 * (R[1, 0] - R[0, 1]) / a
    else:
        xd, yd, zd = 1.0 + R[0, 0] - (R[1, 1] + R[2, 2]), 1.0 + R[1, 1] - (R[0, 0] + R[2, 2]), 1.0 + R[2, 2] - (R[0, 0] + R[1, 1])
        if xd > 1.0:
            b = 0.5 * np.sqrt(xd); c = 0.25 * (R[0, 1] + R[1, 0]) / b; d = 0.25 * (R[0, 2] + R[2, 0]) / b; a = 0.25 * (R[2, 1] - R[1, 2]) / b
        elif yd > 1.0:
            c = 0.5 * np.sqrt(yd); b = 0.25 * (R[0, 1] + R[1, 0]) / c; d = 0.25 * (R[1, 2] + R[2, 1]) / c; a = 0.25 * (R[0, 2] - R[2, 0]) / c
        else:
            d = 0.5 * np.sqrt(zd); b = 0.25 * (R[0, 2] + R[2, 0]) / d; c = 0.25 * (R[1, 2] + R[2, 1]) / d; a = 0.25 * (R[1, 0] - R[0, 1]) / d
        if a < 0.0:
            b, c, d = -b, -c, -d
    return float(b), float(c), float(d), qfac


def read_nifti(path: str, dtype: Optional[np.dtype] = np.float32) -> Image:
    """``itk.imread(path, itk.F)``: returns an ``Image`` in ITK (LPS) conventions, array [z, y, x] (or [c, z, y, x])."""
    with _open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 352:
        raise NiftiError(f"{path}: too short for a NIfTI-1 file")
    endian = "<"
    if struct.unpack("<i", raw[:4])[0] != 348:
        if struct.unpack(">i", raw[:4])[0] != 348:
            raise NiftiError(f"{path}: sizeof_hdr is not 348 in either byte order")
        endian = ">"
    magic = raw[344:348]
    if magic not in (b"n+1\0", b"ni1\0"):
        raise NiftiError(f"{path}: bad magic {magic!r}")
    if magic == b"ni1\0":
        raise NiftiError(f"{path}: two-file NIfTI (.hdr/.img) is not supported")
    dim = struct.unpack(endian + "8h", raw[40:56])
    intent_code, datatype, bitpix = struct.unpack(endian + "3h", raw[68:74])
    pixdim = struct.unpack(endian + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(endian + "3f", raw[108:120])
    qform_code, sform_code = struct.unpack(endian + "2h", raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(endian + "6f", raw[256:280])
    srow = np.array(struct.unpack(endian + "12f", raw[280:328]), dtype=np.float64).reshape(3, 4)
    if datatype not in _DTYPES:
        raise NiftiError(f"{path}: unsupported datatype code {datatype}")
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise NiftiError(f"{path}: bad dim[0] = {ndim}")
    shape_xyz = [max(int(dim[i]), 1) for i in (1, 2, 3)]
    ncomp = int(dim[5]) if ndim >= 5 else 1
    if ndim >= 4 and dim[4] > 1:
        raise NiftiError(f"{path}: time series (dim[4] = {dim[4]}) are outside this package's scope")
    count = shape_xyz[0] * shape_xyz[1] * shape_xyz[2] * ncomp
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(endian)
    off = int(vox_offset) if vox_offset >= 352 else 352
    if len(raw) < off + count * dt.itemsize:
        raise NiftiError(f"{path}: truncated voxel data")
    data = np.frombuffer(raw, dtype=dt, count=count, offset=off)
    arr = data.reshape((ncomp, shape_xyz[2], shape_xyz[1], shape_xyz[0]) if ncomp > 1 else (shape_xyz[2], shape_xyz[1], shape_xyz[0]))
    if slope not in (0.0, 1.0) or (slope == 1.0 and inter != 0.0):
        arr = arr.astype(np.float64) * float(slope) + float(inter)
    arr = np.ascontiguousarray(arr.astype(dtype if dtype is not None else dt.newbyteorder("="), copy=False))

    # ---- orientation: RAS affine -> ITK LPS spacing / origin / direction
    if sform_code > 0:
        M = srow[:, :3]
        spacing = np.linalg.norm(M, axis=0)
        spacing[spacing == 0] = 1.0
        direction_ras, origin_ras = M / spacing, srow[:, 3]
    elif qform_code > 0:
        qfac = -1.0 if pixdim[0] < 0 else 1.0
        direction_ras = _quaternion_to_matrix(qb, qc, qd, qfac)
        spacing = np.array([abs(pixdim[1]) or 1.0, abs(pixdim[2]) or 1.0, abs(pixdim[3]) or 1.0], dtype=np.float64)
        origin_ras = np.array([qx, qy, qz], dtype=np.float64)
    else:
        direction_ras = np.eye(3)
        spacing = np.array([abs(pixdim[1]) or 1.0, abs(pixdim[2]) or 1.0, abs(pixdim[3]) or 1.0], dtype=np.float64)
        origin_ras = np.zeros(3)
    img = Image(arr, spacing, _LPS @ origin_ras, _LPS @ direction_ras)
    img.intent_code = int(intent_code)
    return img


def write_nifti(path: str, image: Image, intent_code: int = 0, description: str = "oai_analysis_2_amd") -> None:
    """``itk.imwrite``: single-file NIfTI-1 (gzip when the name ends in .gz); both qform and sform are written (codes 1)."""
    arr = np.asarray(image.array)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    if arr.dtype.name not in _CODES:
        raise NiftiError(f"cannot store dtype {arr.dtype} in NIfTI-1")
    if arr.ndim == 3:
        nz, ny, nx = arr.shape
        ncomp = 1
    elif arr.ndim == 4:
        ncomp, nz, ny, nx = arr.shape
        intent_code = intent_code or NIFTI_INTENT_VECTOR
    else:
        raise NiftiError("write_nifti takes [z,y,x] or [c,z,y,x] arrays")
    direction_ras = _LPS @ image.direction
    origin_ras = _LPS @ image.origin
    b, c, d, qfac = _matrix_to_quaternion(direction_ras)
    srow = np.concatenate([direction_ras * image.spacing[None, :], origin_ras[:, None]], axis=1).astype(np.float32)
    hdr = bytearray(348)
    struct.pack_into("<i", hdr, 0, 348)
    dim = [5, nx, ny, nz, 1, ncomp, 1, 1] if ncomp > 1 else [3, nx, ny, nz, 1, 1, 1, 1]
    struct.pack_into("<8h", hdr, 40, *dim)
    struct.pack_into("<3h", hdr, 68, int(intent_code), _CODES[arr.dtype.name], arr.dtype.itemsize * 8)
    struct.pack_into("<8f", hdr, 76, qfac, *[float(v) for v in image.spacing], 0.0, 1.0 if ncomp > 1 else 0.0, 0.0, 0.0)
    struct.pack_into("<3f", hdr, 108, 352.0, 1.0, 0.0)
    hdr[123] = 2                                           # xyzt_units: millimetres
    desc = description.encode("ascii", "replace")[:79]
    hdr[148:148 + len(desc)] = desc
    struct.pack_into("<2h", hdr, 252, 1, 1)                # qform_code, sform_code = NIFTI_XFORM_SCANNER_ANAT
    struct.pack_into("<6f", hdr, 256, b, c, d, *[float(v) for v in origin_ras])
    struct.pack_into("<12f", hdr, 280, *srow.reshape(-1).tolist())
    hdr[344:348] = b"n+1\0"
    with _open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(b"\0\0\0\0")                               # no header extensions
        f.write(np.ascontiguousarray(arr).astype(arr.dtype.newbyteorder("<"), copy=False).tobytes())


def write_displacement_nifti(path: str, disp_zyx3: np.ndarray, reference: Image) -> None:
    """Store an ITK displacement field (``create_itk_transform``'s f64 [D,H,W,3] array of xyz vectors, physical units of
    ``reference``'s grid) the way ``itk.imwrite`` stores ``itk.Image[itk.Vector[itk.D,3],3]``.  ITK vectors are LPS, NIfTI
    vectors RAS: x and y components flip sign, as itkNiftiImageIO does for intent 1007."""
    d = np.asarray(disp_zyx3)
    if d.ndim != 4 or d.shape[-1] != 3:
        raise NiftiError("displacement must be [z,y,x,3]")
    comp = np.moveaxis(d, -1, 0).copy()
    comp[0] *= -1.0
    comp[1] *= -1.0
    write_nifti(path, Image(comp, reference.spacing, reference.origin, reference.direction), NIFTI_INTENT_VECTOR)


def read_displacement_nifti(path: str) -> Tuple[np.ndarray, Image]:
    img = read_nifti(path, dtype=None)
    if img.array.ndim != 4 or img.array.shape[0] != 3:
        raise NiftiError(f"{path}: not a 3-component vector image")
    comp = img.array.astype(np.float64)
    comp[0] *= -1.0
    comp[1] *= -1.0
    return np.ascontiguousarray(np.moveaxis(comp, 0, -1)), Image(comp[0], img.spacing, img.origin, img.direction)
