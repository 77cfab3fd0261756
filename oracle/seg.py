"""Oracle: overlap-tiled 3D U-Net cartilage segmentation (CPU, torch fp32 / numpy).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  A functional restatement of

* ``UNet.forward``                      oai_analysis/segmentation/networks.py:38-149
* ``Partition.__call__`` / ``assemble`` oai_analysis/segmentation/image_transforms.py:388-519
* ``Segmenter3DInPatchClassWise.segment`` oai_analysis/segmentation/segmenter.py:100-131

Pinned against the reference itself: tests/golden/make_golden.py imports the
reference classes (with an ``itk`` shim) and stores their outputs;
tests/test_oracle_golden.py checks this file against those vectors.

Everything works on a plain ``state_dict`` (reference key names, Appendix B of
SURVEY.md) so that no ``nn.Module`` of the reference is needed at run time.
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# layer tables: name -> kind, in the order UNet.forward runs them (networks.py:109-149)
ENCODER = ["ec0", "ec1", "ec2", "ec3", "ec4", "ec5", "ec6", "ec7"]          # Conv3d k3 s1 p1  (:43-50)
UPCONV = ["dc9", "dc6", "dc3"]                                              # ConvTranspose3d k2 s2 (:56,59,62)
DECONV3 = ["dc8", "dc7", "dc5", "dc4", "dc2", "dc1"]                        # ConvTranspose3d k3 s1 p1 (:57-64)
BN_EPS = 1e-5                                                               # nn.BatchNorm3d default


def _block(x: torch.Tensor, sd: Dict[str, torch.Tensor], name: str) -> torch.Tensor:
    """One ``nn.Sequential(conv|convT, [BatchNorm3d], ReLU)`` block (networks.py:80-107)."""
    w = sd[f"{name}.0.weight"]
    b = sd.get(f"{name}.0.bias")
    if name in ENCODER:
        y = F.conv3d(x, w, b, stride=1, padding=1)
    elif name in UPCONV:
        y = F.conv_transpose3d(x, w, b, stride=2, padding=0)
    elif name in DECONV3:
        y = F.conv_transpose3d(x, w, b, stride=1, padding=1)
    else:  # pragma: no cover
        raise KeyError(name)
    if f"{name}.1.running_mean" in sd:  # eval-mode BatchNorm3d (segmenter.py:61 -> model.eval())
        y = F.batch_norm(y, sd[f"{name}.1.running_mean"], sd[f"{name}.1.running_var"],
                         sd[f"{name}.1.weight"], sd[f"{name}.1.bias"], training=False, eps=BN_EPS)
    return F.relu(y)


@torch.no_grad()
def unet_forward(x: torch.Tensor, sd: Dict[str, torch.Tensor], return_all: bool = False):
    """logits[B,n_classes,D,H,W] = UNet(x[B,1,D,H,W]); networks.py:109-149."""
    acts = {}
    e0 = _block(x, sd, "ec0")
    syn0 = _block(e0, sd, "ec1")
    e1 = F.max_pool3d(syn0, 2)
    e2 = _block(e1, sd, "ec2")
    syn1 = _block(e2, sd, "ec3")
    e3 = F.max_pool3d(syn1, 2)
    e4 = _block(e3, sd, "ec4")
    syn2 = _block(e4, sd, "ec5")
    e5 = F.max_pool3d(syn2, 2)
    e6 = _block(e5, sd, "ec6")
    e7 = _block(e6, sd, "ec7")
    u9 = _block(e7, sd, "dc9")
    d8 = _block(torch.cat((u9, syn2), 1), sd, "dc8")        # concat order (up, skip)  :127
    d7 = _block(d8, sd, "dc7")
    u6 = _block(d7, sd, "dc6")
    d5 = _block(torch.cat((u6, syn1), 1), sd, "dc5")        # :134
    d4 = _block(d5, sd, "dc4")
    u3 = _block(d4, sd, "dc3")
    d2 = _block(torch.cat((u3, syn0), 1), sd, "dc2")        # :141
    d1 = _block(d2, sd, "dc1")
    d0 = F.conv3d(d1, sd["dc0.weight"], sd.get("dc0.bias"))  # 1x1x1 head, no ReLU   :66,148
    if return_all:
        acts.update(e0=e0, syn0=syn0, e1=e1, e2=e2, syn1=syn1, e3=e3, e4=e4, syn2=syn2, e5=e5, e6=e6,
                    e7=e7, u9=u9, d8=d8, d7=d7, u6=u6, d5=d5, d4=d4, u3=u3, d2=d2, d1=d1, d0=d0)
        return d0, acts
    return d0


# ----------------------------------------------------------------------------------------------
# Partition / assemble


def tile_geometry(image_size_zyx: Sequence[int], patch_size_xyz: Sequence[int],
                  overlap_size_xyz: Sequence[int]) -> dict:
    """Tile grid of ``Partition`` (image_transforms.py:388-415).

    ``patch_size`` / ``overlap_size`` come in (x,y,z) order and are flipped to numpy
    (z,y,x) order (:389-391).  Returns everything in (z,y,x) order.
    """
    size = np.asarray(image_size_zyx, dtype=np.int64)
    tile = np.asarray(patch_size_xyz, dtype=np.int64)[::-1].copy()
    ovl = np.asarray(overlap_size_xyz, dtype=np.int64)[::-1].copy()
    eff = tile - 2 * ovl                                            # :407
    if np.any(eff <= 0):
        raise ValueError("overlap too large for the tile")
    grid = np.ceil(size / eff).astype(np.int64)                     # :408
    extra = eff * grid + 2 * ovl - size                             # :409 (total padding per axis)
    pad_lo = ovl.copy()
    pad_hi = extra - ovl                                            # :411-414
    return dict(size=size, tile=tile, overlap=ovl, effective=eff, grid=grid,
                pad_lo=pad_lo, pad_hi=pad_hi, n_tiles=int(np.prod(grid)))


def partition(vol_zyx: np.ndarray, patch_size_xyz, overlap_size_xyz) -> Tuple[np.ndarray, dict]:
    """Reflect-pad and cut ``N x 1 x d x h x w`` tiles, z-major tile order (:411-446)."""
    g = tile_geometry(vol_zyx.shape, patch_size_xyz, overlap_size_xyz)
    padded = np.pad(vol_zyx, tuple((int(l), int(h)) for l, h in zip(g["pad_lo"], g["pad_hi"])), mode="reflect")
    e, t = g["effective"], g["tile"]
    tiles = []
    for i in range(g["grid"][0]):
        for j in range(g["grid"][1]):
            for k in range(g["grid"][2]):
                tiles.append(padded[i * e[0]:i * e[0] + t[0], j * e[1]:j * e[1] + t[1], k * e[2]:k * e[2] + t[2]])
    return np.stack(tiles, 0)[:, None], g


def assemble(tiles: np.ndarray, g: dict, crop_size_xyz=None) -> np.ndarray:
    """Stitch tile centres, trim, zero the outer frame; returns float64 (:492-513).

    ``crop_size`` is indexed (x,y,z): ``[2]`` is z, ``[0]`` is applied to numpy axis 1 and
    ``[1]`` to numpy axis 2 -- exactly as the reference does (:511-512).
    """
    e, t, o, grid = g["effective"], g["tile"], g["overlap"], g["grid"]
    out = np.zeros(tuple(int(v) for v in e * grid))                  # float64, :493
    n = 0
    for i in range(grid[0]):
        for j in range(grid[1]):
            for k in range(grid[2]):
                out[i * e[0]:(i + 1) * e[0], j * e[1]:(j + 1) * e[1], k * e[2]:(k + 1) * e[2]] = \
                    tiles[n][o[0]:t[0] - o[0], o[1]:t[1] - o[1], o[2]:t[2] - o[2]]
                n += 1
    s = g["size"]
    out = out[:s[0], :s[1], :s[2]]
    if crop_size_xyz:
        c = crop_size_xyz
        framed = np.zeros(out.shape)
        framed[c[2]:-c[2], c[0]:-c[0], c[1]:-c[1]] = out[c[2]:-c[2], c[0]:-c[0], c[1]:-c[1]]
        out = framed
    return out


@torch.no_grad()
def segment(vol_zyx: np.ndarray, sd: Dict[str, torch.Tensor], patch_size_xyz=(128, 128, 32),
            overlap_size_xyz=(16, 16, 8), batch_size: int = 4, output_prob: bool = True,
            return_logits: bool = False):
    """(FC, TC) float64 maps exactly as ``Segmenter3DInPatchClassWise.segment`` (segmenter.py:100-131)."""
    tiles, g = partition(np.asarray(vol_zyx), patch_size_xyz, overlap_size_xyz)
    tiles_t = torch.from_numpy(np.ascontiguousarray(tiles))
    outs = []
    for s in range(0, tiles_t.shape[0], batch_size):                 # :108-119
        outs.append(unet_forward(tiles_t[s:s + batch_size], sd))
    logits = torch.cat(outs, 0)
    pred = torch.sigmoid(logits)                                     # :121
    if not output_prob:
        pred = pred > 0.5                                            # :123-124
    fc = assemble(pred[:, 0].numpy(), g, crop_size_xyz=overlap_size_xyz)   # :126-129
    tc = assemble(pred[:, 1].numpy(), g, crop_size_xyz=overlap_size_xyz)
    if return_logits:
        lf = assemble(logits[:, 0].numpy(), g, crop_size_xyz=None)
        lt = assemble(logits[:, 1].numpy(), g, crop_size_xyz=None)
        return fc, tc, lf, lt
    return fc, tc


# ----------------------------------------------------------------------------------------------
# bookkeeping used by DESIGN.md / bench.py roofline accounting

UNET_LAYERS = [  # name, kind, cin, cout, level
    ("ec0", "c3", 1, 32, 0), ("ec1", "c3", 32, 64, 0), ("ec2", "c3", 64, 64, 1), ("ec3", "c3", 64, 128, 1),
    ("ec4", "c3", 128, 128, 2), ("ec5", "c3", 128, 256, 2), ("ec6", "c3", 256, 256, 3), ("ec7", "c3", 256, 512, 3),
    ("dc9", "up", 512, 512, 2), ("dc8", "c3", 768, 256, 2), ("dc7", "c3", 256, 256, 2),
    ("dc6", "up", 256, 256, 1), ("dc5", "c3", 384, 128, 1), ("dc4", "c3", 128, 128, 1),
    ("dc3", "up", 128, 128, 0), ("dc2", "c3", 192, 64, 0), ("dc1", "c3", 64, 64, 0), ("dc0", "c1", 64, 2, 0),
]


def unet_flops(tile_zyx=(32, 128, 128), overlap_zyx=(8, 16, 16), trimmed=False) -> float:
    """2*MACs of one tile; ``trimmed`` applies the bit-identical dead-output trim of Appendix B.1."""
    need = trim_regions(tile_zyx, overlap_zyx) if trimmed else None
    total = 0.0
    for name, kind, cin, cout, lvl in UNET_LAYERS:
        dims = [t >> lvl for t in tile_zyx]
        if need is not None:
            lo, hi = need[name]
            dims = [h - l for l, h in zip(lo, hi)]
        vox = dims[0] * dims[1] * dims[2]
        taps = {"c3": 27, "up": 1, "c1": 1}[kind]
        total += 2.0 * vox * taps * cin * cout
    return total


def trim_regions(tile_zyx=(32, 128, 128), overlap_zyx=(8, 16, 16)) -> Dict[str, Tuple[list, list]]:
    """Output box [lo,hi) every layer must produce so the kept centre is bit-identical (App. B.1)."""
    full = {l: [t >> l for t in tile_zyx] for l in range(4)}
    need: Dict[str, Tuple[list, list]] = {}

    def grow(box, lvl):   # what a k3 p1 conv reads to produce `box`
        lo, hi = box
        return [max(0, a - 1) for a in lo], [min(f, b + 1) for b, f in zip(hi, full[lvl])]

    def halve(box):       # what a k2 s2 up-conv reads to produce `box`
        lo, hi = box
        return [a // 2 for a in lo], [-(-b // 2) for b in hi]

    box = ([o for o in overlap_zyx], [t - o for t, o in zip(tile_zyx, overlap_zyx)])
    need["dc0"] = box
    need["dc1"] = box
    need["dc2"] = grow(need["dc1"], 0)
    need["dc3"] = grow(need["dc2"], 0)
    need["dc4"] = halve(need["dc3"])
    need["dc5"] = grow(need["dc4"], 1)
    need["dc6"] = grow(need["dc5"], 1)
    need["dc7"] = halve(need["dc6"])
    need["dc8"] = grow(need["dc7"], 2)
    for name, kind, cin, cout, lvl in UNET_LAYERS:
        if name not in need:   # encoder and dc9 stay full: the bottleneck sees the whole tile
            need[name] = ([0, 0, 0], list(full[lvl]))
    return need
