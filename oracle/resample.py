"""Oracle: pull a probability map onto the atlas grid through phi (CPU, numpy float64).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

*** PARITY UNPINNED ***  In the reference this step is ITK C++:
``itk.resample_image_filter(prob, transform=phi_AB, interpolator=LinearInterpolateImageFunction,
size/spacing/direction/origin of image_B)`` at test/test_all.py:42-52 and
oai_analysis/dask_processing.py:95-111, with ``phi_AB`` the CompositeTransform built by
``icon_registration.itk_wrapper.create_itk_transform``.  ITK is not installed here and the
reference's asserts after this step are commented out (test/test_all.py:69-70).  Restated from
ITK's documented behaviour:

* ``ResampleImageFilter``: for every output index, physical point p (output = image_B geometry),
  q = transform(p), continuous index of q in the input image; if inside the buffer
  ([-0.5, n-0.5) per axis) the interpolated value, else the default pixel value 0.
* ``LinearInterpolateImageFunction``: trilinear with neighbours clamped to the buffer.
* ``CompositeTransform`` [to_network_space, DisplacementFieldTransform, from_network_space] applied
  back to front: B-physical -> network index space -> + displacement -> A-physical.
* ``DisplacementFieldTransform``: linear interpolation of the vector field (neighbours clamped) when
  the point is inside the field's buffer, identity outside.
"""
from __future__ import annotations

import numpy as np

from .icon import network_affine


def _trilinear_clamped(vol: np.ndarray, ix: np.ndarray, iy: np.ndarray, iz: np.ndarray) -> np.ndarray:
    """vol[z,y,x] (optionally trailing channel) sampled at continuous indices, clamped to the buffer."""
    nz, ny, nx = vol.shape[:3]
    cx = np.clip(ix, 0.0, nx - 1.0)
    cy = np.clip(iy, 0.0, ny - 1.0)
    cz = np.clip(iz, 0.0, nz - 1.0)
    x0 = np.floor(cx).astype(np.int64); y0 = np.floor(cy).astype(np.int64); z0 = np.floor(cz).astype(np.int64)
    x1 = np.minimum(x0 + 1, nx - 1); y1 = np.minimum(y0 + 1, ny - 1); z1 = np.minimum(z0 + 1, nz - 1)
    fx = cx - x0; fy = cy - y0; fz = cz - z0
    if vol.ndim == 4:
        fx, fy, fz = fx[..., None], fy[..., None], fz[..., None]
    v = vol.astype(np.float64, copy=False)
    c00 = v[z0, y0, x0] * (1 - fx) + v[z0, y0, x1] * fx
    c01 = v[z0, y1, x0] * (1 - fx) + v[z0, y1, x1] * fx
    c10 = v[z1, y0, x0] * (1 - fx) + v[z1, y0, x1] * fx
    c11 = v[z1, y1, x0] * (1 - fx) + v[z1, y1, x1] * fx
    c0 = c00 * (1 - fy) + c01 * fy
    c1 = c10 * (1 - fy) + c11 * fy
    return c0 * (1 - fz) + c1 * fz


def resample_through_phi(prob_A: np.ndarray, disp_itk: np.ndarray, meta_A, meta_B, chunk_z: int = 16) -> np.ndarray:
    """warped[z,y,x] on image_B's grid = prob_A(phi_AB(p)); float64.

    ``disp_itk``: float64 [D,H,W,3] from :func:`oracle.icon.displacement_itk` (xyz components,
    network voxel units).  ``meta_A`` / ``meta_B``: objects with spacing, origin, direction, size_xyz
    (``oai_analysis_2_amd.image.Image``).
    """
    net_shape = disp_itk.shape[:3]
    M_A, c_net, c_A = network_affine(meta_A.spacing, meta_A.origin, meta_A.direction, meta_A.size_xyz, net_shape)
    M_B, _, c_B = network_affine(meta_B.spacing, meta_B.origin, meta_B.direction, meta_B.size_xyz, net_shape)
    M_B_inv = np.linalg.inv(M_B)
    P_B = meta_B.direction @ np.diag(meta_B.spacing)                 # B index -> physical
    P_A_inv = np.linalg.inv(meta_A.direction @ np.diag(meta_A.spacing))  # A physical -> index
    nxB, nyB, nzB = (int(v) for v in meta_B.size_xyz)
    nzA, nyA, nxA = prob_A.shape
    out = np.zeros((nzB, nyB, nxB), dtype=np.float64)
    Dn, Hn, Wn = net_shape
    for z0 in range(0, nzB, chunk_z):
        z1 = min(nzB, z0 + chunk_z)
        zz, yy, xx = np.meshgrid(np.arange(z0, z1, dtype=np.float64), np.arange(nyB, dtype=np.float64),
                                 np.arange(nxB, dtype=np.float64), indexing="ij")
        idx = np.stack([xx, yy, zz], -1)                              # [.,.,.,3] xyz
        p = idx @ P_B.T + meta_B.origin                               # B physical
        x = (p - c_B) @ M_B_inv.T + c_net                             # network index space (from_network_space)
        inside = ((x[..., 0] >= -0.5) & (x[..., 0] < Wn - 0.5) & (x[..., 1] >= -0.5) & (x[..., 1] < Hn - 0.5) &
                  (x[..., 2] >= -0.5) & (x[..., 2] < Dn - 0.5))
        d = _trilinear_clamped(disp_itk, x[..., 0], x[..., 1], x[..., 2])
        x2 = x + np.where(inside[..., None], d, 0.0)                  # DisplacementFieldTransform
        q = (x2 - c_net) @ M_A.T + c_A                                # A physical (to_network_space)
        ia = (q - meta_A.origin) @ P_A_inv.T                          # A continuous index
        ok = ((ia[..., 0] >= -0.5) & (ia[..., 0] < nxA - 0.5) & (ia[..., 1] >= -0.5) & (ia[..., 1] < nyA - 0.5) &
              (ia[..., 2] >= -0.5) & (ia[..., 2] < nzA - 0.5))
        val = _trilinear_clamped(prob_A, ia[..., 0], ia[..., 1], ia[..., 2])
        out[z0:z1] = np.where(ok, val, 0.0)
    return out
