"""Cohort driver: stream many volumes through one GPU (or one GPU per rank) with the weights resident.

Replaces the reference's Dask task graph for this path (dask_processing.py:46-189), which rebuilds the
segmenter and the ICON model inside every task (:77, :170) and moves pickled ITK images between workers.
Here the engines are built once per process; volume i+1 is uploaded on a side stream (pinned host buffer,
PCIe Gen5: 94 MB in ~1.5 ms) while volume i computes, and results are copied back asynchronously.
"""
from __future__ import annotations

from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .image import Image, as_image
from .pipeline import VolumePipeline, VolumeResult


class CohortRunner:
    """Three things overlap per volume: the compute of volume i (main stream), the host staging + H2D of volume i+1 and the D2H
    of volume i-1's results (copy stream, persistent pinned buffers).  The host never blocks before the next compute is queued."""

    def __init__(self, pipeline: VolumePipeline, keep_on_device: bool = False):
        self.pipe = pipeline
        self.keep_on_device = keep_on_device
        self.copy_stream = torch.cuda.Stream(device=pipeline.unet.device)
        self._pin_in: List[Optional[torch.Tensor]] = [None, None]       # double-buffered pinned upload staging
        self._pin_ev: List[Optional[torch.cuda.Event]] = [None, None]  # H2D out of each staging buffer has finished
        self._pin_out: dict = {}

    def _upload(self, img: Image, slot: int) -> Tuple[torch.Tensor, torch.cuda.Event]:
        src = torch.from_numpy(np.ascontiguousarray(img.array, dtype=np.float32))
        buf = self._pin_in[slot]
        if buf is None or buf.shape != src.shape:
            buf = self._pin_in[slot] = torch.empty(src.shape, dtype=torch.float32).pin_memory()
        if self._pin_ev[slot] is not None:
            self._pin_ev[slot].synchronize()                              # the previous upload out of this buffer is done
        buf.copy_(src)                                                    # host memcpy into page-locked memory (no re-pinning per volume)
        with torch.cuda.stream(self.copy_stream):
            dev = buf.to(self.pipe.unet.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._pin_ev[slot] = ev
        return dev, ev

    def _download(self, res: VolumeResult, done: torch.cuda.Event) -> Optional[VolumeResult]:
        """D2H of one volume's results on the copy stream (after `done`), through reusable pinned buffers.  The fp16 range flag
        of the volume rides along (4 bytes); None = it overflowed and the results must not be used."""
        self.copy_stream.wait_event(done)
        outs = []
        with torch.cuda.stream(self.copy_stream):
            names = ("fc", "tc", "phi", "fc_atlas", "tc_atlas") + (("overflow",) if res.overflow is not None else ())
            for name in names:
                t = getattr(res, name)
                key = (name, tuple(t.shape), t.dtype)
                if key not in self._pin_out:
                    self._pin_out[key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
                self._pin_out[key].copy_(t, non_blocking=True)
                t.record_stream(self.copy_stream)
                outs.append(self._pin_out[key])
        self.copy_stream.synchronize()
        if res.overflow is not None and int(outs[5][0]):
            return None
        return VolumeResult(*(o.clone() for o in outs[:5]), repeated_f32=res.repeated_f32)   # the pinned buffers are reused by the next volume

    def _finish(self, pending) -> VolumeResult:
        """Results of a queued volume; a volume whose fp16x3 segmentation left fp16's range is repeated in exact fp32 here
        (the check is lazy -- at download time -- so the overlap of the normal case is kept; never silent)."""
        i, res, done, dev, img = pending
        if self.keep_on_device:
            done.synchronize()
            if res.overflow is not None and int(res.overflow.item()):
                res = self.pipe.rerun_f32(dev, img)
                torch.cuda.current_stream().synchronize()
            return res
        out = self._download(res, done)
        if out is None:
            res = self.pipe.rerun_f32(dev, img)
            done = torch.cuda.Event()
            done.record()
            out = self._download(res, done)
        return out

    def run(self, images: Sequence, rank: int = 0, world: int = 1, queue=None) -> Iterator[Tuple[int, VolumeResult]]:
        """Yield (index, result) for the volumes this worker processes, in its processing order.  ``queue`` (a
        ``parallel.VolumeQueue`` shared by all ranks) assigns volumes dynamically -- the worker claims one volume ahead, so that its
        upload overlaps the current compute; without a queue the static split index % world == rank is used."""
        if queue is not None:
            order = iter(queue)
        else:
            order = iter([i for i in range(len(images)) if i % world == rank])
        cur = next(order, None)
        if cur is None:
            return
        # every image is fetched from ``images`` exactly ONCE (a lazy sequence -- dask_processing._Lazy -- reads and normalises the file
        # in __getitem__) and travels with its upload: (device tensor, upload event, host image)
        img = as_image(images[cur])
        nxt = (*self._upload(img, 0), img)
        pending = None                                                    # (index, device results, completion event, ...) of the previous volume
        k = 0
        while cur is not None:
            dev, ev, img = nxt
            torch.cuda.current_stream().wait_event(ev)
            dev.record_stream(torch.cuda.current_stream())                # allocated on the copy stream, read by the compute stream
            res = self.pipe.run(dev, img, check=False)                    # queued, not waited for; the range flag is read at download time
            done = torch.cuda.Event()
            done.record()
            following = next(order, None)                                 # claimed now: its host staging + H2D run behind this volume's compute
            if following is not None:
                img_next = as_image(images[following])
                nxt = (*self._upload(img_next, (k + 1) & 1), img_next)
            if pending is not None:
                yield pending[0], self._finish(pending)
            pending = (cur, res, done, dev, img)
            cur, k = following, k + 1
        yield pending[0], self._finish(pending)
