"""``Partition`` with the reference's interface (image_transforms.py:371-519), tiles as addressing.

On the HIP path no tile tensor is ever materialised for ``segment`` (the first conv gathers straight
from the resident volume with the same reflect-pad index math).  ``__call__`` and ``assemble`` are kept
for callers that use them directly; they produce/consume device tensors.
"""
from __future__ import annotations

import numpy as np
import torch

from ..image import Image, as_image
from .engine import tile_grid


class Partition(object):
    def __init__(self, tile_size, overlap_size, padding_mode="reflect", mode="eval"):
        if padding_mode != "reflect":
            raise NotImplementedError("only padding_mode='reflect' (the reference's pred configuration)")
        self.tile_size = np.flipud(np.asarray(tile_size))          # (x,y,z) -> (z,y,x)   :389-391
        self.overlap_size = np.flipud(np.asarray(overlap_size))
        self.padding_mode, self.mode = padding_mode, mode

    def geometry(self, image_size_zyx):
        self.image_size = np.asarray(image_size_zyx)
        eff, grid, n = tile_grid(image_size_zyx, self.tile_size, self.overlap_size)
        self.effective_size, self.tiles_grid_size = np.asarray(eff), np.asarray(grid)
        return eff, grid, n

    def __call__(self, sample):
        """{'image': Image|array} -> {'image': torch [N,1,d,h,w] on the current HIP device} (:395-455)."""
        img = as_image(sample["image"])
        self.image, self.name = img, sample.get("name", "")
        vol = torch.from_numpy(np.ascontiguousarray(img.array, dtype=np.float32)).cuda()
        eff, grid, n = self.geometry(vol.shape)
        t, o = [int(v) for v in self.tile_size], [int(v) for v in self.overlap_size]
        from .. import _lib
        lib = _lib.load()
        tiles = torch.empty((n, 1, *t), dtype=torch.float32, device=vol.device)
        with torch.cuda.device(vol.device):          # the gather is a HIP kernel (oai_partition_tiles), not torch indexing
            _lib.check(lib.oai_partition_tiles(vol.data_ptr(), *[int(v) for v in vol.shape], _lib.int3(t), _lib.int3(o), 0, n,
                                               tiles.data_ptr(), torch.cuda.current_stream().cuda_stream), "oai_partition_tiles")
        sample = dict(sample)
        sample["image"] = tiles
        return sample

    def assemble(self, tiles, is_vote=False, if_itk=True, crop_size=None, data_type=None):
        """Partition.assemble (image_transforms.py:457-519), non-vote branch: kept tile centres -> image, trimmed to the image
        size, the outer frame of ``crop_size`` zeroed (``crop_size`` indexed like the reference: [2] is z, [0] lands on numpy
        axis 1, [1] on axis 2; a zero component zeroes everything, :509-513).  Runs on the device (oai_stitch_blocks);
        returns float64 like the reference unless ``data_type`` is given."""
        from .. import _lib
        lib = _lib.load()
        if is_vote:
            return self._assemble_vote(lib, tiles, if_itk, crop_size, data_type)
        t = torch.as_tensor(np.asarray(tiles) if not isinstance(tiles, torch.Tensor) else tiles).to(torch.float32)
        if t.dim() == 5:
            t = t[:, 0]
        tz, ty, tx = (int(v) for v in self.tile_size)
        oz, oy, ox = (int(v) for v in self.overlap_size)
        D, H, W = (int(v) for v in self.image_size)
        eff, grid, n = tile_grid((D, H, W), (tz, ty, tx), (oz, oy, ox))
        if t.shape[0] != n or tuple(t.shape[1:]) != (tz, ty, tx):
            raise ValueError(f"assemble needs {n} tiles of {tz}x{ty}x{tx}, got {tuple(t.shape)}")
        blocks = t[:, oz:tz - oz, oy:ty - oy, ox:tx - ox].contiguous().cuda()            # [N][ez][ey][ex], one class
        maps = torch.empty((1, D, H, W), dtype=torch.float32, device=blocks.device)
        crop = _lib.int3((int(crop_size[2]), int(crop_size[0]), int(crop_size[1]))) if crop_size is not None and len(crop_size) else None
        with torch.cuda.device(blocks.device):
            _lib.check(lib.oai_stitch_blocks(blocks.data_ptr(), 1, D, H, W, _lib.int3((tz, ty, tx)), _lib.int3((oz, oy, ox)), crop,
                                             maps.data_ptr(), torch.cuda.current_stream().cuda_stream), "oai_stitch_blocks")
        out = maps[0].cpu().numpy().astype(data_type if data_type else np.float64)
        return self._finish(out, if_itk)

    def _finish(self, out, if_itk):
        if if_itk:
            img = Image(out)
            if getattr(self, "image", None) is not None:
                img.CopyInformation(self.image)
            return img
        return out

    def _assemble_vote(self, lib, tiles, if_itk, crop_size, data_type):
        """The vote branch (image_transforms.py:466-484) on the device: labels must be the integers 0..L-1 (the reference indexes its
        vote array with the label value); uint8 result, float64 once ``crop_size`` is applied (np.zeros canvas, :509-513)."""
        from .. import _lib
        t = torch.as_tensor(np.asarray(tiles) if not isinstance(tiles, torch.Tensor) else tiles)
        if t.dim() == 5:
            t = t[:, 0]
        if t.is_floating_point():
            raise IndexError("only integers are valid label indices (assemble(is_vote=True) indexes the vote array with the label value)")
        tz, ty, tx = (int(v) for v in self.tile_size)
        oz, oy, ox = (int(v) for v in self.overlap_size)
        D, H, W = (int(v) for v in self.image_size)
        eff, grid, n = tile_grid((D, H, W), (tz, ty, tx), (oz, oy, ox))
        if t.shape[0] != n or tuple(t.shape[1:]) != (tz, ty, tx):
            raise ValueError(f"assemble needs {n} tiles of {tz}x{ty}x{tx}, got {tuple(t.shape)}")
        classes = np.unique(t.cpu().numpy())                              # label_class = np.unique(tiles), :468 (host metadata)
        nlab = int(classes.size)
        if int(classes.min()) < 0 or int(classes.max()) >= nlab:
            raise IndexError(f"label {int(classes.max())} is out of bounds for the vote array of {nlab} label planes")
        lab = t.to(torch.int32).contiguous().cuda()
        if nlab > 16:
            raise NotImplementedError("more than 16 label classes")
        out = torch.empty((D, H, W), dtype=torch.uint8, device=lab.device)
        with torch.cuda.device(lab.device):
            _lib.check(lib.oai_assemble_vote(lab.data_ptr(), nlab, D, H, W, _lib.int3((tz, ty, tx)), _lib.int3((oz, oy, ox)),
                                             out.data_ptr(), torch.cuda.current_stream().cuda_stream), "oai_assemble_vote")
        res = out.cpu().numpy()
        if data_type:
            res = res.astype(data_type)
        if crop_size is not None and len(crop_size):
            cz, cy, cx = int(crop_size[2]), int(crop_size[0]), int(crop_size[1])
            canvas = np.zeros(res.shape)                                   # float64, like the reference
            if cz and cy and cx:                                           # [c:-c] with c == 0 is empty
                canvas[cz:-cz, cy:-cy, cx:-cx] = res[cz:-cz, cy:-cy, cx:-cx]
            res = canvas
        return self._finish(res, if_itk)
