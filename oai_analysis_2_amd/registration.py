"""ICON atlas registration on MI355X behind the reference's ``ICON_Registration`` surface.

Mirrors oai_analysis/registration.py:18-27: ``ICON_Registration().register(fixed, moving)`` returns
phi_fixed_moving -- the map that pulls a *fixed*-space (patient) image onto the *moving* (atlas)
grid, i.e. ``resample(patient_prob, transform=phi, reference=atlas)`` (test/test_all.py:42-58).

The arithmetic of ``icon_registration.itk_wrapper.register_pair`` (resize -> three tallUNet2s with
warps/composes -> dense phi -> displacement in network-voxel units + two centring affines) runs as
hand-written HIP kernels through liboai_hip.so; see oracle/icon.py for the restated algorithm.
ITK is optional: the result is a :class:`DisplacementTransform` (numpy displacement field + affines,
the exact content of the ITK CompositeTransform) with ``to_itk()`` when ``itk`` is importable.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from .image import Image, as_image

NET_SHAPE = (80, 192, 192)   # OAI_knees_gradICON_model input_shape[2:]


def network_affine(img: Image, net_shape_dhw: Sequence[int]):
    """``itk_wrapper.resampling_transform(image, shape)``: p_phys = M (x_net - c_net) + c_img (xyz)."""
    size = img.size_xyz.astype(np.float64)
    shape = np.asarray(net_shape_dhw[::-1], np.float64)
    M = img.direction @ np.diag(img.spacing * (size / shape))
    c_net = (shape - 1.0) / 2.0
    c_img = img.origin + img.direction @ (img.spacing * (size - 1.0) / 2.0)
    return M, c_net, c_img


def resample_affines(image_A: Image, image_B: Image, net_shape_dhw: Sequence[int]):
    """Host-side fp64 composition of the two affine legs of phi_AB for ``oai_resample_through_disp``.

    b_index_to_net : B index -> physical_B -> network index space   (from_network_space)
    net_to_a_index : network index space -> physical_A -> A continuous index   (to_network_space)
    """
    M_A, c_net, c_A = network_affine(image_A, net_shape_dhw)
    M_B, _, c_B = network_affine(image_B, net_shape_dhw)
    P_B, o_B = image_B.index_to_physical_affine()
    P_A, o_A = image_A.index_to_physical_affine()
    M_B_inv = np.linalg.inv(M_B)
    A1 = M_B_inv @ P_B
    b1 = M_B_inv @ (o_B - c_B) + c_net
    P_A_inv = np.linalg.inv(P_A)
    A2 = P_A_inv @ M_A
    b2 = P_A_inv @ (c_A - M_A @ c_net - o_A)
    return (A1, b1), (A2, b2)


@dataclass
class DisplacementTransform:
    """What ``create_itk_transform`` builds: B-physical -> A-physical through a network-space field."""
    displacement: np.ndarray            # float64 [D,H,W,3], xyz components, network-voxel units
    image_A: Image                      # metadata only (array may be dropped)
    image_B: Image
    phi: Optional[np.ndarray] = None    # float32 [3,D,H,W] dense map in [0,1] units (network grid)

    @property
    def net_shape(self):
        return self.displacement.shape[:3]

    def affines(self):
        return resample_affines(self.image_A, self.image_B, self.net_shape)

    def to_itk(self):  # pragma: no cover - itk is absent in this environment
        import itk
        dim = 3
        tr = itk.DisplacementFieldTransform[(itk.D, dim)].New()
        tr.SetDisplacementField(itk.image_from_array(np.ascontiguousarray(self.displacement), is_vector=True))

        def affine(img):
            M, c_net, c_img = network_affine(img, self.net_shape)
            t = itk.CenteredAffineTransform[itk.D, 3].New()
            t.SetCenter([float(v) for v in c_net])
            t.SetOffset([float(v) for v in (c_img - c_net)])
            t.SetMatrix(itk.matrix_from_array(np.ascontiguousarray(M)))
            return t

        comp = itk.CompositeTransform[itk.D, dim].New()
        comp.PrependTransform(affine(self.image_B).GetInverseTransform())
        comp.PrependTransform(tr)
        comp.PrependTransform(affine(self.image_A))
        return comp
