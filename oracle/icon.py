"""Oracle: ICON gradICON atlas registration forward (CPU, torch fp32).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

*** PARITY UNPINNED ***  The arithmetic lives in the third-party package
``icon_registration==1.1.2`` (pinned at /root/reference/pyproject.toml:35), which is neither
vendored in /root/reference nor installed in this image, and the reference's own tests at this
boundary assert nothing (test/test_all.py:72-81 and :88-99 have their asserts commented out or
absent).  This file restates the package's published algorithm; the reference call sites it is
anchored on are oai_analysis/registration.py:20 (``OAI_knees_gradICON_model``) and :25
(``itk_wrapper.register_pair``), and oai_analysis/dask_processing.py:77,85.

Published structure restated (module.function of icon_registration 1.1.2):

* ``networks.tallUNet2`` / ``UNet2.forward``       -> :func:`tall_unet2`
* ``network_wrappers.FunctionFromVectorField``,
  ``TwoStepRegistration``, ``DownsampleRegistration`` -> :func:`regis_net_direction`
* ``mermaidlite.compute_warped_image_multiNC`` (``scale_map`` + xyz reorder + ``grid_sample``)
                                                    -> :func:`sample_at`
* ``mermaidlite.identity_map_multiN``               -> :func:`identity_map`
* ``itk_wrapper.register_pair``                     -> :func:`register_pair_arrays`
* ``itk_wrapper.create_itk_transform`` / ``resampling_transform`` -> :func:`displacement_itk`,
                                                       :func:`network_affine`

At inference the similarity and gradICON losses of ``GradientICON.forward`` are computed and
thrown away; they do not influence phi and are not restated.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

DOWN = [2, 16, 32, 64, 256, 512]
UP_OUT = [16, 32, 64, 128, 256]
NET_SHAPE = (80, 192, 192)          # OAI_knees_gradICON_model: input_shape = [1,1,40*2,96*2,96*2]
BN_EPS = 1e-5
LEAKY = 0.01                         # F.leaky_relu default slope

U1 = "netPhi.net.netPhi.net."       # the three-step tree of SURVEY Appendix A: low-res step 1
U2 = "netPhi.net.netPsi.net."       # low-res step 2
U3 = "netPsi.net."                   # full-res step


# Two points of the restatement rest on recollection of the un-vendored package and can be recalled differently (SURVEY App. A;
# VERDICT r2 missing #4): whether ``UNet2.forward`` applies ``batchNorms[depth]`` (the ModuleList exists either way, so the keys are
# in the state_dict) and on which side ``pad_or_crop`` puts the zero channels.  Both are switches here and in the library
# (oai_icon_create: bn_* == NULL; oai_icon_set_option "pad_front"), tested in all four combinations; the defaults are SURVEY's.
# Round 6: apply_bn = None lets the checkpoint decide (infer_apply_bn below): BatchNorm statistics that moved, or a batch counter
# above zero, are proof that UNet2.forward calls ``self.batchNorms[depth]`` -- they only change when the module is called in
# training mode -- and pristine tensors make the switch irrelevant (x / sqrt(1 + 1e-5) per level).
OPTIONS = {"apply_bn": None, "pad_front": True}


def infer_apply_bn(sd) -> bool:
    """True iff the checkpoint's ``batchNorms.*`` entries show that the layers were called while it was trained: a
    ``num_batches_tracked`` above zero, or gamma / beta / running_mean / running_var off their constructor values (1, 0, 0, 1)."""
    for k, v in sd.items():
        if ".batchNorms." not in k and not k.startswith("batchNorms."):
            continue
        t = torch.as_tensor(v)
        if t.numel() == 0:
            continue
        if k.endswith("num_batches_tracked"):
            if int(t.max()) > 0:
                return True
            continue
        want = 1.0 if (k.endswith(".weight") or k.endswith("running_var")) else 0.0
        if float((t.float() - want).abs().max()) > 1e-6:
            return True
    return False


def _pad_or_crop_channels(x: torch.Tensor, c: int, pad_front: bool = True) -> torch.Tensor:
    """``networks.pad_or_crop``: keep the first ``c`` channels, or zero-pad (in front, or behind) up to ``c``."""
    y = x[:, :c]
    if x.shape[1] < c:
        n = c - x.shape[1]
        y = F.pad(y, (0, 0, 0, 0, 0, 0, n, 0) if pad_front else (0, 0, 0, 0, 0, 0, 0, n))
    return y


@torch.no_grad()
def tall_unet2(a: torch.Tensor, b: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str = "", apply_bn=None, pad_front=None) -> torch.Tensor:
    """``UNet2(5, [[2,16,32,64,256,512],[16,32,64,128,256]], 3).forward(a, b)`` -> [B,3,D,H,W].  ``apply_bn`` / ``pad_front``:
    None = the module-level OPTIONS."""
    apply_bn = OPTIONS["apply_bn"] if apply_bn is None else apply_bn
    if apply_bn is None:
        apply_bn = infer_apply_bn(sd)            # the whole checkpoint decides, not one U-Net's slice of it
    pad_front = OPTIONS["pad_front"] if pad_front is None else pad_front
    x = torch.cat([a, b], 1)
    skips = []
    for d in range(5):
        skips.append(x)
        y = F.conv3d(F.leaky_relu(x, LEAKY), sd[f"{prefix}downConvs.{d}.weight"], sd[f"{prefix}downConvs.{d}.bias"],
                     stride=2, padding=1)
        x = y + _pad_or_crop_channels(F.avg_pool3d(x, 2, ceil_mode=True), y.shape[1], pad_front)
    for d in reversed(range(5)):
        y = F.conv_transpose3d(F.leaky_relu(x, LEAKY), sd[f"{prefix}upConvs.{d}.weight"], sd[f"{prefix}upConvs.{d}.bias"],
                               stride=2, padding=1)
        x = y + F.interpolate(_pad_or_crop_channels(x, y.shape[1], pad_front), scale_factor=2, mode="trilinear", align_corners=False)
        if apply_bn:
            x = F.batch_norm(x, sd[f"{prefix}batchNorms.{d}.running_mean"], sd[f"{prefix}batchNorms.{d}.running_var"],
                             sd[f"{prefix}batchNorms.{d}.weight"], sd[f"{prefix}batchNorms.{d}.bias"], training=False, eps=BN_EPS)
        s = skips[d]
        x = x[:, :, :s.shape[2], :s.shape[3], :s.shape[4]]
        x = torch.cat([x, s], 1)
    x = F.conv3d(x, sd[f"{prefix}lastConv.weight"], sd[f"{prefix}lastConv.bias"], padding=1)
    return x / 10


def identity_map(shape_dhw: Sequence[int]) -> torch.Tensor:
    """[1,3,D,H,W] map, channel d = index_d * 1/(n_d-1) in [0,1] (float32 of the float64 product)."""
    D, H, W = shape_dhw
    grids = np.mgrid[0:D, 0:H, 0:W].astype(np.float64)
    for d, n in enumerate((D, H, W)):
        grids[d] *= 1.0 / (n - 1)
    return torch.from_numpy(grids.astype(np.float32))[None]


def sample_at(src: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """``compute_warped_image_multiNC(src, coords, spacing=1/(shape-1), spline_order=1)``.

    coords in [0,1]: g = 2*coords - 1, channels (d,h,w) -> grid (x,y,z), trilinear, border, align_corners=True.
    """
    g = coords * 2.0 - 1.0
    grid = torch.stack([g[:, 2], g[:, 1], g[:, 0]], dim=-1)
    return F.grid_sample(src, grid, mode="bilinear", padding_mode="border", align_corners=True)


_UNET_ATTRS = {"downConvs", "upConvs", "batchNorms", "lastConv", "residues"}
_BUFFERS = {"identity_map", "spacing", "num_batches_tracked", "_extra_state"}


def _children(sd, prefix):
    return sorted({k[len(prefix):].split(".", 1)[0] for k in sd
                   if k.startswith(prefix) and k.rsplit(".", 1)[-1] not in _BUFFERS})


def _apply(links, coords, tagged_identity: bool):
    """A closure of the package applied to a coordinate map: ``links`` are the displacement tensors of the FFVFs in application
    order (for TwoStep's ``lambda x: phi(psi(x))``: psi's links, then phi's).  ``FunctionFromVectorField.forward``'s
    ``transform``: ``coords + d`` if coords is the tagged identity map and has d's shape, else ``coords + sample(d, coords)``;
    the result of either is an ordinary tensor (the tag does not propagate)."""
    for d in links:
        if tagged_identity and coords.shape == d.shape:
            coords = coords + d
        else:
            coords = coords + sample_at(d, coords)
        tagged_identity = False
    return coords


def forward_tree(sd: Dict[str, torch.Tensor], prefix: str, A: torch.Tensor, B: torch.Tensor, trace=None):
    """``module.forward(A, B)`` of the wrapper module whose parameters live under ``prefix`` -- the module type is read off the
    names of its children exactly as ``nn.Module.state_dict`` spells them -- returned as the list of its closure's links.

    * ``TwoStepRegistration`` (children netPhi, netPsi): ``phi = netPhi(A, B)``;
      ``psi = netPsi(as_function(A)(phi(self.identity_map)), B)`` with ``self.identity_map`` tagged isIdentity at the top of
      forward; returns ``lambda x: phi(psi(x))``.
    * ``DownsampleRegistration`` (child net = another wrapper): ``net(avg_pool(A, 2, ceil_mode=True), avg_pool(B, ...))``.
    * ``FunctionFromVectorField`` (child net = a UNet2): ``d = net(A, B)``.
    """
    kids = _children(sd, prefix)
    if kids == ["netPhi", "netPsi"]:
        phi = forward_tree(sd, prefix + "netPhi.", A, B, trace)
        ident = identity_map(A.shape[2:])
        A_w = sample_at(A, _apply(phi, ident, True))
        if trace is not None:
            trace.append(("warp", prefix, A_w))
        psi = forward_tree(sd, prefix + "netPsi.", A_w, B, trace)
        return psi + phi
    if kids == ["net"]:
        grand = _children(sd, prefix + "net.")
        if grand and set(grand) <= _UNET_ATTRS:
            d = tall_unet2(A, B, sd, prefix + "net.")
            if trace is not None:
                trace.append(("unet", prefix + "net.", d))
            return [d]
        return forward_tree(sd, prefix + "net.", F.avg_pool3d(A, 2, ceil_mode=True), F.avg_pool3d(B, 2, ceil_mode=True), trace)
    raise KeyError(f"not a registration wrapper under '{prefix}': {kids[:4]}")


@torch.no_grad()
def regis_net_direction(A: torch.Tensor, B: torch.Tensor, sd: Dict[str, torch.Tensor], return_all: bool = False):
    """phi_AB(identity) = ``regis_net(A, B)(identity_map)`` for whatever wrapper tree ``sd``'s keys spell, e.g.
    TwoStep(Downsample(TwoStep(FFVF(u1),FFVF(u2))), FFVF(u3)) or the four-step form with one more FFVF around it.

    A, B: [1,1,D,H,W] network-resolution images.  Returns the dense map [1,3,D,H,W] in [0,1] units
    (``GradientICON.forward`` tags ``identity_map.isIdentity``, ``register_pair`` evaluates ``model.phi_AB(model.identity_map)``).
    """
    trace = [] if return_all else None
    links = forward_tree(sd, "", A, B, trace)
    phi = _apply(links, identity_map(A.shape[2:]), True)
    if return_all:
        return phi, trace
    return phi


@torch.no_grad()
def regis_net_direction_3step_unrolled(A: torch.Tensor, B: torch.Tensor, sd: Dict[str, torch.Tensor]):
    """The three-step tree written out by hand (rounds 1-3's oracle): a cross-check of the recursion above."""
    id_h = identity_map(A.shape[2:])
    a = F.avg_pool3d(A, 2, ceil_mode=True)                         # DownsampleRegistration.forward
    b = F.avg_pool3d(B, 2, ceil_mode=True)
    id_l = identity_map(a.shape[2:])
    d1 = tall_unet2(a, b, sd, U1)                                   # FFVF(u1)
    a_w = sample_at(a, id_l + d1)                                   # tagged identity, same shape -> shortcut
    d2 = tall_unet2(a_w, b, sd, U2)                                 # FFVF(u2)
    c1 = id_h + sample_at(d2, id_h)                                 # outer TwoStep: id_h has another shape -> sampled path
    c2 = c1 + sample_at(d1, c1)
    A_w = sample_at(A, c2)
    d3 = tall_unet2(A_w, B, sd, U3)                                 # FFVF(u3)
    c3 = id_h + d3                                                  # tagged identity, same shape -> shortcut
    c4 = c3 + sample_at(d2, c3)
    return c4 + sample_at(d1, c4)


@torch.no_grad()
def register_pair_arrays(image_A: np.ndarray, image_B: np.ndarray, sd: Dict[str, torch.Tensor],
                         net_shape: Sequence[int] = NET_SHAPE, both: bool = True):
    """``itk_wrapper.register_pair`` up to (not including) the ITK transform objects."""
    A = torch.from_numpy(np.asarray(image_A, dtype=np.float32))[None, None]
    B = torch.from_numpy(np.asarray(image_B, dtype=np.float32))[None, None]
    A_r = F.interpolate(A, size=tuple(net_shape), mode="trilinear", align_corners=False)
    B_r = F.interpolate(B, size=tuple(net_shape), mode="trilinear", align_corners=False)
    phi_AB = regis_net_direction(A_r, B_r, sd)
    phi_BA = regis_net_direction(B_r, A_r, sd) if both else None
    return phi_AB, phi_BA


def displacement_itk(phi: torch.Tensor) -> np.ndarray:
    """``create_itk_transform``'s vector image: float64 [D,H,W,3], components (x,y,z), network-voxel units."""
    ident = identity_map(phi.shape[2:])
    disp = (phi - ident)[0]
    scale = torch.tensor([n - 1 for n in phi.shape[2:]], dtype=torch.float32)[:, None, None, None]
    disp = disp * scale
    return disp.double().numpy()[::-1].transpose(1, 2, 3, 0).copy()


def network_affine(spacing_xyz, origin_xyz, direction, size_xyz, net_shape_dhw=NET_SHAPE):
    """``resampling_transform(image, shape)``: network index space -> image physical space.

    Returns (M, c_net, c_img) with  p_phys = M @ (x_net - c_net) + c_img  (all xyz, float64).
    """
    spacing = np.asarray(spacing_xyz, np.float64)
    size = np.asarray(size_xyz, np.float64)
    shape = np.asarray(net_shape_dhw[::-1], np.float64)              # network size in xyz
    direction = np.asarray(direction, np.float64).reshape(3, 3)
    M = direction @ np.diag(spacing * (size / shape))
    c_net = (shape - 1.0) / 2.0
    c_img = np.asarray(origin_xyz, np.float64) + direction @ (spacing * (size - 1.0) / 2.0)
    return M, c_net, c_img
