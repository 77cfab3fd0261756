"""Tensor-level wrappers over the C ABI (torch is plumbing: device memory + the current stream).

Every function takes contiguous fp32 ``torch`` tensors on the HIP device, passes ``data_ptr()``
and the current stream to liboai_hip.so, and returns the output tensor.  No torch compute op is
used on the product path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _on_tensor_device(fn):
    """Run ``fn`` with the device of its first tensor argument current: the C ABI launches on the current HIP device and takes the
    current stream, so a tensor on another GPU of the process must switch both."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if torch.is_tensor(a) and a.is_cuda:
                with torch.cuda.device(a.device):
                    return fn(*args, **kwargs)
            if isinstance(a, (list, tuple)) and a and torch.is_tensor(a[0]) and a[0].is_cuda:
                with torch.cuda.device(a[0].device):
                    return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return wrapper


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.OaiError(f"{name} must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.OaiError(f"{name} must be {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def warp_set_option(name: str, value: int) -> None:
    """Process-wide tuning option of the warp kernels (include/oai_hip.h: oai_warp_set_option): "brick" 0|1 -- grid_sample3d / compose through
    the LDS-staged brick kernel; bit-identical outputs."""
    _lib.check(_lib.load().oai_warp_set_option(name.encode(), int(value)), "oai_warp_set_option")


@_on_tensor_device
def grid_sample3d(src: torch.Tensor, coords: Optional[torch.Tensor], out_shape: Optional[Sequence[int]] = None) -> torch.Tensor:
    """src [C,d,h,w], coords [3,D,H,W] in [0,1] (None = identity of out_shape) -> [C,D,H,W]."""
    lib = _lib.load()
    src = _chk(src, "src")
    Cn, d, h, w = src.shape
    if coords is not None:
        coords = _chk(coords, "coords")
        D, H, W = coords.shape[1:]
    else:
        D, H, W = out_shape
    out = torch.empty((Cn, D, H, W), dtype=torch.float32, device=src.device)
    _lib.check(lib.oai_grid_sample3d(src.data_ptr(), Cn, d, h, w, coords.data_ptr() if coords is not None else None,
                                     D, H, W, out.data_ptr(), _stream()), "oai_grid_sample3d")
    return out


@_on_tensor_device
def compose(disp: torch.Tensor, coords: Optional[torch.Tensor], out_shape: Optional[Sequence[int]] = None,
            shortcut: bool = True) -> torch.Tensor:
    """coords + sample(disp, coords); coords None = identity map of out_shape (or of disp's grid)."""
    lib = _lib.load()
    disp = _chk(disp, "disp")
    _, d, h, w = disp.shape
    if coords is not None:
        coords = _chk(coords, "coords")
        D, H, W = coords.shape[1:]
    else:
        D, H, W = out_shape if out_shape is not None else (d, h, w)
    out = torch.empty((3, D, H, W), dtype=torch.float32, device=disp.device)
    _lib.check(lib.oai_compose(disp.data_ptr(), d, h, w, coords.data_ptr() if coords is not None else None,
                               D, H, W, int(shortcut), out.data_ptr(), _stream()), "oai_compose")
    return out


@_on_tensor_device
def warp_chain(out_shape: Sequence[int], fields: Sequence[torch.Tensor] = (), start: Optional[torch.Tensor] = None,
               image: Optional[torch.Tensor] = None) -> torch.Tensor:
    """c = identity(out_shape) [+ start]; c = c + sample(f, c) for f in fields (<= 8 = OAI_WARP_CHAIN_MAX_FIELDS); returns sample(image, c) [D,H,W] when an
    image [d,h,w] is given, else c [3,D,H,W].  One launch, bit-identical to the compose / grid_sample3d calls it replaces."""
    lib = _lib.load()
    D, H, W = (int(v) for v in out_shape)
    fields = [_chk(f, "field") for f in fields]
    if len(fields) > 8 or any(f.dim() != 4 or f.shape[0] != 3 for f in fields):
        raise ValueError("at most eight fields, each [3,d,h,w]")
    dev = (fields[0] if fields else start if start is not None else image).device
    if start is not None:
        start = _chk(start, "start")
        if tuple(start.shape) != (3, D, H, W):
            raise ValueError("start must be [3,D,H,W] on the output grid")
    if image is not None:
        image = _chk(image, "image")
        if image.dim() != 3:
            raise ValueError("image must be [d,h,w]")
    ptrs = (C.c_void_p * max(1, len(fields)))(*[f.data_ptr() for f in fields])
    dims = (C.c_int * max(3, 3 * len(fields)))(*[int(v) for f in fields for v in f.shape[1:]])
    out = torch.empty((D, H, W) if image is not None else (3, D, H, W), dtype=torch.float32, device=dev)
    idims = tuple(image.shape) if image is not None else (0, 0, 0)
    with torch.cuda.device(dev):
        _lib.check(lib.oai_warp_chain(start.data_ptr() if start is not None else None, D, H, W, len(fields), ptrs, dims,
                                      image.data_ptr() if image is not None else None, *idims, out.data_ptr(), _stream()), "oai_warp_chain")
    return out


@_on_tensor_device
def avgpool2(x: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    x = _chk(x, "x")
    Cn, D, H, W = x.shape
    out = torch.empty((Cn, (D + 1) // 2, (H + 1) // 2, (W + 1) // 2), dtype=torch.float32, device=x.device)
    _lib.check(lib.oai_avgpool2_3d(x.data_ptr(), Cn, D, H, W, out.data_ptr(), _stream()), "oai_avgpool2_3d")
    return out


@_on_tensor_device
def resize_trilinear(x: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    lib = _lib.load()
    x = _chk(x, "x")
    Cn, d, h, w = x.shape
    D, H, W = (int(v) for v in size)
    out = torch.empty((Cn, D, H, W), dtype=torch.float32, device=x.device)
    _lib.check(lib.oai_resize_trilinear(x.data_ptr(), Cn, d, h, w, out.data_ptr(), D, H, W, _stream()), "oai_resize_trilinear")
    return out


@_on_tensor_device
def phi_to_itk_displacement(phi: torch.Tensor) -> torch.Tensor:
    """phi [3,D,H,W] -> float64 [D,H,W,3] (xyz components, network voxel units)."""
    lib = _lib.load()
    phi = _chk(phi, "phi")
    _, D, H, W = phi.shape
    out = torch.empty((D, H, W, 3), dtype=torch.float64, device=phi.device)
    _lib.check(lib.oai_phi_to_itk_displacement(phi.data_ptr(), D, H, W, out.data_ptr(), _stream()), "oai_phi_to_itk_displacement")
    return out


def make_affine(A: np.ndarray, b: np.ndarray) -> _lib.Affine:
    a = _lib.Affine()
    a.A[:] = [float(v) for v in np.asarray(A, np.float64).reshape(9)]
    a.b[:] = [float(v) for v in np.asarray(b, np.float64).reshape(3)]
    return a


@_on_tensor_device
def resample_through_disp(prob: torch.Tensor, disp: torch.Tensor, b_index_to_net, net_to_a_index,
                          out_shape_zyx: Sequence[int]) -> torch.Tensor:
    lib = _lib.load()
    prob = _chk(prob, "prob")
    disp = _chk(disp, "disp", torch.float64)
    nzA, nyA, nxA = prob.shape
    Dn, Hn, Wn, _ = disp.shape
    nzB, nyB, nxB = (int(v) for v in out_shape_zyx)
    out = torch.empty((nzB, nyB, nxB), dtype=torch.float32, device=prob.device)
    a1, a2 = make_affine(*b_index_to_net), make_affine(*net_to_a_index)
    _lib.check(lib.oai_resample_through_disp(prob.data_ptr(), nzA, nyA, nxA, disp.data_ptr(), Dn, Hn, Wn,
                                             C.byref(a1), C.byref(a2), out.data_ptr(), nzB, nyB, nxB, _stream()),
               "oai_resample_through_disp")
    return out


@_on_tensor_device
def resample_maps_through_phi(maps: torch.Tensor, phi: torch.Tensor, b_index_to_net, net_to_a_index,
                              out_shape_zyx: Sequence[int]) -> torch.Tensor:
    """maps [n,zA,yA,xA] (n <= 4) pulled through the dense map phi [3,D,H,W] onto a grid of ``out_shape_zyx``: one launch,
    bit-identical to ``phi_to_itk_displacement`` + ``resample_through_disp`` per map."""
    lib = _lib.load()
    maps = _chk(maps, "maps")
    phi = _chk(phi, "phi")
    if maps.dim() != 4 or phi.dim() != 4 or phi.shape[0] != 3:
        raise ValueError("maps must be [n,z,y,x] and phi [3,D,H,W]")
    n, nzA, nyA, nxA = maps.shape
    _, Dn, Hn, Wn = phi.shape
    nzB, nyB, nxB = (int(v) for v in out_shape_zyx)
    out = torch.empty((n, nzB, nyB, nxB), dtype=torch.float32, device=maps.device)
    a1, a2 = make_affine(*b_index_to_net), make_affine(*net_to_a_index)
    with torch.cuda.device(maps.device):
        _lib.check(lib.oai_resample_maps_through_phi(maps.data_ptr(), n, nzA, nyA, nxA, phi.data_ptr(), Dn, Hn, Wn,
                                                     C.byref(a1), C.byref(a2), out.data_ptr(), nzB, nyB, nxB, _stream()),
                   "oai_resample_maps_through_phi")
    return out


@_on_tensor_device
def image_normalize(vol: torch.Tensor, window_min_perc: float = 0.1, window_max_perc: float = 99.9,
                    output_min: float = 0.0, output_max: float = 1.0, return_window: bool = False):
    """``image_normalize`` of oai_analysis/dask_processing.py:10-26 on the device (fp32 image)."""
    lib = _lib.load()
    vol = _chk(vol, "vol")
    out = torch.empty_like(vol)
    ws = torch.empty(int(lib.oai_image_normalize_workspace_bytes()), dtype=torch.uint8, device=vol.device)
    win = torch.empty(2, dtype=torch.float32, device=vol.device)
    _lib.check(lib.oai_image_normalize(vol.data_ptr(), vol.numel(), float(window_min_perc), float(window_max_perc),
                                       float(output_min), float(output_max), out.data_ptr(), win.data_ptr(),
                                       ws.data_ptr(), ws.numel(), _stream()), "oai_image_normalize")
    return (out, win) if return_window else out
