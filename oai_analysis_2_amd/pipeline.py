"""The per-volume hot path, device resident end to end:

    DESS volume (fp32, [z,y,x]) --segment--> FC/TC probability maps --+
                                --register--> phi (atlas -> patient) --+--> FC/TC on the atlas grid

i.e. AnalysisObject.segment + AnalysisObject.register + the two ``deform_probmap`` calls of the
reference's pipeline (test/test_all.py:54-58, dask_processing.py:46-125), without any host round trip.
Used by bench.py, the cohort driver and the multi-GPU sharding.
"""
from __future__ import annotations

from dataclasses import dataclass
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
    overflow: Optional[torch.Tensor] = None    # int32[1] on the device: the fp16 range flag of THIS volume's segmentation
    #                                            (None with exact arithmetic).  Non-zero = the maps are invalid: repeat in fp32.
    repeated_f32: bool = False   # the fp16x3 run overflowed and these are the results of the fp32 repeat


class VolumePipeline:
    """Range guard of the default fp16x3 arithmetic (|activation| <= 65504): every entry point that returns results either
    checks the flag of its own volume and repeats that volume in exact fp32 (``check=True``, the default: one stream
    synchronisation AFTER everything of the volume is queued), or hands the flag snapshot back in ``VolumeResult.overflow`` for a
    caller that pipelines volumes and checks at download time (``check=False``: CohortRunner, bench.py).  Never silent."""

    def __init__(self, unet: UNetEngine, icon: IconEngine, atlas: Image, tile_zyx=TILE_ZYX, overlap_zyx=OVERLAP_ZYX,
                 crop_zyx=CROP_ZYX, batch: Optional[int] = None):
        self.unet, self.icon, self.atlas = unet, icon, atlas
        self.tile_zyx, self.overlap_zyx, self.crop_zyx, self.batch = tuple(tile_zyx), tuple(overlap_zyx), tuple(crop_zyx), batch
        self.atlas_dev = torch.from_numpy(np.ascontiguousarray(atlas.array, dtype=np.float32)).to(unet.device)
        self._atlas_net = None
        self._side = None
        self.overlap_registration = True          # registration underneath the segmentation (+1.5 %); False serialises

    # ---- stages ---------------------------------------------------------------------------------------------------------------
    def segment(self, vol: torch.Tensor, out_mode: int = 0, tile_range: Optional[Tuple[int, int]] = None):
        blocks = self.unet.segment_tiles(vol, self.tile_zyx, self.overlap_zyx, tile_range, out_mode, self.batch, self.crop_zyx)
        if tile_range is not None:
            return blocks
        return self.unet.stitch(blocks, vol.shape, self.tile_zyx, self.overlap_zyx, self.crop_zyx)

    def _flag_snapshot(self) -> Optional[torch.Tensor]:
        """Queue (no sync) a copy-and-clear of the fp16 range flag behind the segment calls queued so far."""
        if self.unet.precision != "fp16x3":
            return None
        flag = torch.zeros(1, dtype=torch.int32, device=self.unet.device)
        self.unet.range_overflow_snapshot(flag)
        return flag

    def register(self, vol: torch.Tensor) -> torch.Tensor:
        """phi_AB with A = patient volume (fixed), B = atlas (moving): registration.py:22-27."""
        A = ops.resize_trilinear(vol[None], self.icon.net_shape)[0]
        if self._atlas_net is None:      # the atlas is the same for every volume: resize it once
            self._atlas_net = ops.resize_trilinear(self.atlas_dev[None], self.icon.net_shape)[0]
        return self.icon.phi(A, self._atlas_net)

    def resample(self, maps: torch.Tensor, phi: torch.Tensor, meta_A: Image, z_range: Optional[Tuple[int, int]] = None):
        """Both maps pulled onto the atlas grid through phi (ONE launch reads the displacement once for all maps);
        ``z_range`` = only atlas slices [z0, z1) (the z-slab shard of SURVEY 8e).  Returns [n_maps, z, y, x]."""
        b2n, n2a = resample_affines(meta_A, self.atlas, self.icon.net_shape)
        shape = tuple(self.atlas.array.shape)
        if z_range is not None:
            z0, z1 = int(z_range[0]), int(z_range[1])
            if not 0 <= z0 <= z1 <= shape[0]:
                raise ValueError("z_range outside the atlas grid")
            A1, b1 = b2n
            b2n = (A1, b1 + A1[:, 2] * float(z0))          # index_B = index_slab + (0, 0, z0) in ITK's x,y,z order
            shape = (z1 - z0, shape[1], shape[2])
        return ops.resample_maps_through_phi(maps, phi, b2n, n2a, shape)

    # ---- one volume, one GPU --------------------------------------------------------------------------------------------------
    def run(self, vol: torch.Tensor, meta_A: Image, check: bool = True) -> VolumeResult:
        res = self._run_overlapped(vol, meta_A) if self.overlap_registration else self._run_serial(vol, meta_A)
        if check and res.overflow is not None:
            raised = bool(int(res.overflow.item()))
            self.unet.note_volume_flag(raised)              # (a calibration FILE that keeps missing the data is dropped after three volumes in a row)
            if raised:
                return self.rerun_f32(vol, meta_A)
        return res

    def rerun_f32(self, vol: torch.Tensor, meta_A: Image, sharded_group="none") -> VolumeResult:
        """The volume left fp16x3's calibrated range window (range flag: overflow or a layer far quieter than at calibration): repeat
        it with exact fp32 MFMA arithmetic (what Segmenter3DInPatchClassWise does)."""
        print("WARNING: activations outside the fp16x3 range window, repeating the volume in fp32")
        prev = self.unet.precision
        self.unet.set_precision("f32")
        try:
            res = self.run(vol, meta_A, check=False) if sharded_group == "none" else self.run_sharded(vol, meta_A, sharded_group, check=False)
        finally:
            self.unet.set_precision(prev)
        res.repeated_f32 = True
        return res

    def _run_serial(self, vol: torch.Tensor, meta_A: Image) -> VolumeResult:
        maps = self.segment(vol)
        flag = self._flag_snapshot()
        phi = self.register(vol)
        atlas_maps = self.resample(maps, phi, meta_A)
        return VolumeResult(maps[0], maps[1], phi, atlas_maps[0], atlas_maps[1], flag)

    def _run_overlapped(self, vol: torch.Tensor, meta_A: Image) -> VolumeResult:
        """Registration needs only the image, not its segmentation: its small, launch- and latency-bound kernels (a few dozen
        workgroups at the deep ICON levels) run on a side stream underneath the segmentation's MFMA kernels; the resample joins."""
        main = torch.cuda.current_stream()
        if self._side is None or self._side.priority != main.priority:
            self._side = torch.cuda.Stream(device=self.unet.device, priority=main.priority)      # (the compute stream's own priority: under a high-priority caller -- CohortRunner -- a default-priority side stream would starve)
        self._side.wait_stream(main)                                    # vol (and the atlas) are ready
        with torch.cuda.stream(self._side):
            phi = self.register(vol)
            phi.record_stream(main)
        maps = self.segment(vol)
        flag = self._flag_snapshot()
        main.wait_stream(self._side)
        atlas_maps = self.resample(maps, phi, meta_A)
        return VolumeResult(maps[0], maps[1], phi, atlas_maps[0], atlas_maps[1], flag)

    # ---- one volume, all ranks of a group (single-volume latency mode) ---------------------------------------------
    def segment_sharded(self, vol: torch.Tensor, group=None) -> torch.Tensor:
        """One volume split over the ranks of ``group`` at tile granularity: every rank runs the U-Net on
        its contiguous tile range, ONE all_gather (RCCL) gives every rank all kept-centre blocks, then stitch."""
        from . import parallel
        _, _, n_tiles = tile_grid(vol.shape, self.tile_zyx, self.overlap_zyx)
        costs = self.unet.tile_costs(vol.shape, self.tile_zyx, self.overlap_zyx, self.crop_zyx)     # border tiles are cheaper: balance the work
        eff, _, _ = tile_grid(vol.shape, self.tile_zyx, self.overlap_zyx)
        # the kernels write this rank's blocks straight into its slot of the gather buffer, the collective runs in place, and the stitch
        # reads the (ragged: 19-23 tiles per rank) buffer through a table: no copy on either side of the all_gather
        gathered = parallel.segment_tile_sharded(
            lambda rng, out: self.unet.segment_tiles(vol, self.tile_zyx, self.overlap_zyx, rng, 0, self.batch, self.crop_zyx, out=out),
            n_tiles, group, costs, block_shape=(self.unet.n_classes, *eff), dtype=torch.float32, device=self.unet.device)
        return self.unet.stitch(gathered, vol.shape, self.tile_zyx, self.overlap_zyx, self.crop_zyx)

    def run_sharded(self, vol: Optional[torch.Tensor], meta_A: Image, group=None, src: int = 0, check: bool = True) -> VolumeResult:
        """``vol`` is needed on rank ``src`` only (others may pass None): broadcast -> tile-sharded segmentation + all_gather ->
        registration replicated (0.1 TFLOP: cheaper than communicating) -> both resamples sharded by atlas z-slab + all_gather.
        The fp16 range flag is all_reduce(MAX)ed: either every rank keeps the result or every rank repeats in fp32."""
        from . import parallel
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        vol = parallel.broadcast_volume(vol, meta_A.array.shape, self.unet.device, src, group)
        # The replicated registration needs only the image: like _run_overlapped, it goes on the side stream UNDERNEATH the sharded
        # segmentation and joins before the resample.  At 8 ranks the segmentation is ~21 ms per rank, so 4.85 ms of ICON + warp
        # kernels in front of or behind it would be ~20 % of the single-volume latency.
        main = torch.cuda.current_stream()
        phi = None
        if self.overlap_registration:
            if self._side is None or self._side.priority != main.priority:
                self._side = torch.cuda.Stream(device=self.unet.device, priority=main.priority)      # (the compute stream's own priority: under a high-priority caller -- CohortRunner -- a default-priority side stream would starve)
            self._side.wait_stream(main)                                # the broadcast volume is ready
            with torch.cuda.stream(self._side):
                phi = self.register(vol)
                phi.record_stream(main)
        maps = self.segment_sharded(vol, group)
        # the range flag of the WHOLE volume: the raw state (overflow bit + per-layer maxima) is MAX-reduced over the ranks and the
        # LOW bit evaluated from that -- a rank whose tile range is all quiet background must not force every rank into the fp32
        # repeat on its own subset's census; after the reduction every rank holds the same verdict
        flag = None
        if self.unet.precision == "fp16x3":
            state = torch.zeros(self.unet.RANGE_STATE_WORDS, dtype=torch.int32, device=self.unet.device)
            self.unet.range_state_snapshot(state)
            parallel.any_rank(state, group)
            flag = self.unet.range_flag_from_state(state)
        if phi is None:
            phi = self.register(vol)
        else:
            main.wait_stream(self._side)
        nz = self.atlas.array.shape[0]
        local = self.resample(maps, phi, meta_A, parallel.slab_range_for_rank(nz, rank, world))
        atlas_maps = parallel.gather_slabs(local, nz, group)
        res = VolumeResult(maps[0], maps[1], phi, atlas_maps[0], atlas_maps[1], flag)
        if check and flag is not None:
            raised = bool(int(flag.item()))                 # (the same verdict on every rank: the state was MAX-reduced)
            self.unet.note_volume_flag(raised)
            if raised:
                return self.rerun_f32(vol, meta_A, sharded_group=group)
        return res
