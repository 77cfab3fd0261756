"""The per-volume hot path, device resident end to end:

    DESS volume (fp32, [z,y,x]) --segment--> FC/TC probability maps --+
                                --register--> phi (atlas -> patient) --+--> FC/TC on the atlas grid

i.e. AnalysisObject.segment + AnalysisObject.register + the two ``deform_probmap`` calls of the
reference's pipeline (test/test_all.py:54-58, dask_processing.py:46-125), without any host round trip.
Used by bench.py, the cohort driver and the multi-GPU sharding.
"""
from __future__ import annotations

from dataclasses import dataclass
import os
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .image import Image
from .registration import IconEngine, resample_affines
from .segmentation.engine import UNetEngine, tile_grid

TILE_ZYX = (32, 128, 128)        # patch_size (128,128,32) in x,y,z (a2)
OVERLAP_ZYX = (8, 16, 16)        # overlap_size (16,16,8) in x,y,z (analysis_object.py:23)
CROP_ZYX = (8, 16, 16)           # assemble's frame: crop_size[2], [0], [1]


@dataclass
class VolumeResult:
    fc: torch.Tensor             # [z,y,x] probability map, patient grid
    tc: torch.Tensor
    phi: torch.Tensor            # [3,D,H,W] dense map, network grid, [0,1] units
    fc_atlas: torch.Tensor       # FC pulled onto the atlas grid through phi
    tc_atlas: torch.Tensor


class VolumePipeline:
    def __init__(self, unet: UNetEngine, icon: IconEngine, atlas: Image, tile_zyx=TILE_ZYX, overlap_zyx=OVERLAP_ZYX,
                 crop_zyx=CROP_ZYX, batch: Optional[int] = None):
        # the conv arithmetic is the engine's (UNetEngine(precision=...)); with "fp16x3" callers that keep results
        # should check unet.range_overflow() once per volume / cohort (Segmenter3DInPatchClassWise does)
        self.unet, self.icon, self.atlas = unet, icon, atlas
        self.tile_zyx, self.overlap_zyx, self.crop_zyx, self.batch = tuple(tile_zyx), tuple(overlap_zyx), tuple(crop_zyx), batch
        self.atlas_dev = torch.from_numpy(np.ascontiguousarray(atlas.array, dtype=np.float32)).to(unet.device)
        self._atlas_net = None
        self._side = None
        self.overlap_registration = os.environ.get("OAI_OVERLAP_REG", "1") == "1"          # registration underneath the segmentation (+1.5 %)

    def segment(self, vol: torch.Tensor, out_mode: int = 0, tile_range: Optional[Tuple[int, int]] = None):
        blocks = self.unet.segment_tiles(vol, self.tile_zyx, self.overlap_zyx, tile_range, out_mode, self.batch, self.crop_zyx)
        if tile_range is not None:
            return blocks
        return self.unet.stitch(blocks, vol.shape, self.tile_zyx, self.overlap_zyx, self.crop_zyx)

    def register(self, vol: torch.Tensor) -> torch.Tensor:
        """phi_AB with A = patient volume (fixed), B = atlas (moving): registration.py:22-27."""
        A = ops.resize_trilinear(vol[None], self.icon.net_shape)[0]
        if self._atlas_net is None:      # the atlas is the same for every volume: resize it once
            self._atlas_net = ops.resize_trilinear(self.atlas_dev[None], self.icon.net_shape)[0]
        return self.icon.phi(A, self._atlas_net)

    def resample(self, maps: torch.Tensor, phi: torch.Tensor, meta_A: Image):
        disp = ops.phi_to_itk_displacement(phi)
        b2n, n2a = resample_affines(meta_A, self.atlas, self.icon.net_shape)
        return [ops.resample_through_disp(maps[c], disp, b2n, n2a, self.atlas.array.shape) for c in range(maps.shape[0])]

    def segment_sharded(self, vol: torch.Tensor, group=None) -> torch.Tensor:
        """One volume split over the ranks of ``group`` at tile granularity: every rank runs the U-Net on
        its contiguous tile range, ONE all_gather (RCCL) gives every rank all kept-centre blocks, then stitch."""
        from . import parallel
        _, _, n_tiles = tile_grid(vol.shape, self.tile_zyx, self.overlap_zyx)
        costs = self.unet.tile_costs(vol.shape, self.tile_zyx, self.overlap_zyx, self.crop_zyx)     # border tiles are cheaper: balance the work
        blocks = parallel.segment_tile_sharded(
            lambda rng: self.unet.segment_tiles(vol, self.tile_zyx, self.overlap_zyx, rng, 0, self.batch, self.crop_zyx), n_tiles, group, costs)
        return self.unet.stitch(blocks, vol.shape, self.tile_zyx, self.overlap_zyx, self.crop_zyx)

    def run_sharded(self, vol: torch.Tensor, meta_A: Image, group=None) -> VolumeResult:
        """Single-volume latency mode: segmentation tile-sharded, registration + resample replicated (0.1 TFLOP)."""
        maps = self.segment_sharded(vol, group)
        phi = self.register(vol)
        fc_a, tc_a = self.resample(maps, phi, meta_A)
        return VolumeResult(maps[0], maps[1], phi, fc_a, tc_a)

    def run(self, vol: torch.Tensor, meta_A: Image) -> VolumeResult:
        if self.overlap_registration:
            return self.run_overlapped(vol, meta_A)
        maps = self.segment(vol)
        phi = self.register(vol)
        fc_a, tc_a = self.resample(maps, phi, meta_A)
        return VolumeResult(maps[0], maps[1], phi, fc_a, tc_a)

    def run_overlapped(self, vol: torch.Tensor, meta_A: Image) -> VolumeResult:
        """Registration needs only the image, not its segmentation: its small, launch- and latency-bound kernels (a few dozen
        workgroups at the deep ICON levels) run on a side stream underneath the segmentation's MFMA kernels; the resample joins."""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.unet.device)
        self._side.wait_stream(main)                                    # vol (and the atlas) are ready
        with torch.cuda.stream(self._side):
            phi = self.register(vol)
            phi.record_stream(main)
        maps = self.segment(vol)
        main.wait_stream(self._side)
        fc_a, tc_a = self.resample(maps, phi, meta_A)
        return VolumeResult(maps[0], maps[1], phi, fc_a, tc_a)
