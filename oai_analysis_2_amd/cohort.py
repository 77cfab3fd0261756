"""Cohort driver: stream many volumes through one GPU (or one GPU per rank) with the weights resident.

Replaces the reference's Dask task graph for this path (dask_processing.py:46-189), which rebuilds the
segmenter and the ICON model inside every task (:77, :170) and moves pickled ITK images between workers.
Here the engines are built once per process; volume i+1 is uploaded on a side stream (pinned host buffer,
PCIe Gen5: 94 MB in ~1.5 ms) while volume i computes, and results are copied back asynchronously.

Round 6: the download of a volume's results is ISSUED BY THE DOWNLOAD WORKER after a host-side wait for the volume's compute -- queued behind the compute on
the GPU, its barrier packet sat at the head of an in-order hardware queue for the whole volume and held back whatever stream shared the queue (_queue_d2h).

Round 5: NO host memcpy on the launch thread.  A volume moves 94 MB pageable -> pinned before its upload and
566 MB pinned -> pageable behind its download (five result tensors); on the thread that queues the kernels those copies
(and the page faults of 566 MB of fresh memory per volume) kept the GPU waiting: 6.03 volumes/s streamed against 7.22 resident.
Both now run on two worker threads (numpy / torch copies release the GIL), over double-buffered pinned staging on either side;
the launch thread only queues work and collects finished results one volume later.
"""
from __future__ import annotations

import queue
import threading
import time
from collections import deque
from concurrent.futures import Future, ThreadPoolExecutor
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .image import Image, as_image
from .pipeline import VolumePipeline, VolumeResult

_RESULT_NAMES = ("fc", "tc", "phi", "fc_atlas", "tc_atlas")


class CohortRunner:
    """Four things overlap per volume: the compute of volume i (main stream, launch thread), the host staging + H2D of volume i+1
    (upload worker, copy stream), the D2H of volume i-1's results (copy stream) and the host copy of volume i-2's results out of
    the pinned buffers into memory the caller owns (download worker).  The launch thread never copies and never waits for a copy
    that is not at least ``LAG`` volumes old.  ``stats`` accumulates what the workers moved (bytes, seconds) for bench.py."""

    # Results are handed out `LAG` volumes behind the compute being queued.  Measured (scripts/bench_cohort.py, GPU-side stamps): the D2H of a
    # volume's 566 MB runs UNDERNEATH the next volume's kernels and then takes ~100 ms, not the 12 ms of an idle PCIe link -- with a lag of one
    # volume the launch thread waited for it and queued the next volume ~4 ms late (0.947 of the resident rate); with two it never waits.
    LAG = 2
    N_OUT_SETS = LAG + 1           # pinned result sets: one per volume whose download may be in flight or being emptied

    def __init__(self, pipeline: VolumePipeline, keep_on_device: bool = False, high_priority_compute: bool = False, board=None, result_pool: int = 0):
        self.pipe = pipeline
        self.keep_on_device = keep_on_device
        # result_pool = n > 0: results are handed out in RECYCLED host tensors -- a ring of n result sets, faulted in once -- instead of freshly allocated
        # ones: nine tenths of the download leg are the page faults of 566 MB of fresh memory per volume (profiles/r06_cohort.md, scripts/host8.py: 58 ms
        # against 15 ms per volume with eight processes on one host).  The caller must be done with a result before n more have been yielded (n >= LAG + 2).
        # 0 (default): every result is memory the caller owns for good, as the reference's per-task returns are (dask_processing.py:170-181).
        if result_pool and result_pool < self.LAG + 2:
            raise ValueError(f"result_pool must be 0 or >= {self.LAG + 2}")
        self.result_pool = int(result_pool)
        self._pool_sets: List[dict] = [dict() for _ in range(self.result_pool)]
        self._pool_next = 0
        # parallel.CalibrationBoard of a multi-rank cohort (process_cohort): when a sidecar's fp16x3 calibration is dropped for not fitting the
        # data, ONE rank recalibrates and every rank takes that outcome -- polled before each volume is queued, never a blocking collective
        self.board = board
        # `high_priority_compute` queues the volumes' kernels on a high-priority stream of the runner's own.  It dates from round 5, when the 2-4 ms between two
        # streamed volumes were blamed on a copy kernel that the compute should pre-empt; round 6 found the real cause (a barrier packet parked in a shared
        # hardware queue: _queue_d2h) and removed it, and found that the runtime's pinned D2H is an SDMA transfer outside the profiler (profiles/r06_cohort.md).
        # The switch is kept for A/B only, NOT the default: it takes the engines off the caller's stream -- a consumer that uses the same pipeline between two
        # results must then synchronise with the runner itself.
        dev_ = pipeline.unet.device
        self.compute_stream = torch.cuda.Stream(device=dev_, priority=torch.cuda.Stream.priority_range()[1]) if high_priority_compute else None
        self.copy_stream = torch.cuda.Stream(device=pipeline.unet.device)      # uploads (and the lazy normalisation of the next volume)
        # downloads on a stream of their own: the D2H of volume i is queued right behind its compute and waits for it (~140 ms); on the
        # upload stream it would hold the H2D of volume i+2 back behind that wait (PCIe is full duplex: the two directions do not compete)
        self.down_stream = torch.cuda.Stream(device=pipeline.unet.device)
        self._pin_in: List[Optional[torch.Tensor]] = [None, None]       # double-buffered pinned upload staging
        self._pin_ev: List[Optional[torch.cuda.Event]] = [None, None]  # H2D out of each staging buffer has finished
        self._pin_out: List[dict] = [dict() for _ in range(self.N_OUT_SETS)]
        self._free_out: "queue.Queue[int]" = queue.Queue()
        for k in range(self.N_OUT_SETS):
            self._free_out.put(k)
        self._up = ThreadPoolExecutor(max_workers=1, thread_name_prefix="oai-upload")
        self._down = ThreadPoolExecutor(max_workers=1, thread_name_prefix="oai-download")
        self._clone = ThreadPoolExecutor(max_workers=5, thread_name_prefix="oai-clone")      # a volume's five result tensors are copied out side by side (one core
        #                                                                                      does ~5-10 GB/s into freshly faulted pages: 566 MB would take most of a volume's compute time)
        self._lock = threading.Lock()
        # what the workers moved, and where the LAUNCH thread spent its time: waiting for an upload, queueing a volume's kernels, issuing its D2H,
        # waiting for a finished result (bench.py: streamed_from_host)
        self.stats = {"stage_bytes": 0, "stage_s": 0.0, "clone_bytes": 0, "clone_s": 0.0, "launch_wait_s": 0.0,
                      "t_upload_wait": 0.0, "t_queue_compute": 0.0, "t_issue_d2h": 0.0, "t_result_wait": 0.0}

    def close(self) -> None:
        self._up.shutdown(wait=True)
        self._down.shutdown(wait=True)
        self._clone.shutdown(wait=True)

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 - interpreter shutdown
            pass

    # ---- upload side (worker thread) --------------------------------------------------------------------------------------------
    def _upload(self, fetch: Callable[[], Image], slot: int) -> Tuple[torch.Tensor, torch.cuda.Event, Image]:
        """Runs on the upload worker: fetch the image (a lazy sequence reads + normalises the file here), stage it in pinned memory,
        queue the H2D on the copy stream.  Returns (device tensor, upload event, host image)."""
        img = as_image(fetch())
        dev_id = self.pipe.unet.device
        with torch.cuda.device(dev_id):
            src = torch.from_numpy(np.ascontiguousarray(img.array, dtype=np.float32))
            buf = self._pin_in[slot]
            if buf is None or buf.shape != src.shape:
                buf = self._pin_in[slot] = torch.empty(src.shape, dtype=torch.float32).pin_memory()
            if self._pin_ev[slot] is not None:
                self._pin_ev[slot].synchronize()                          # the previous upload out of this buffer is done
            t0 = time.perf_counter()
            buf.copy_(src)                                                # host memcpy into page-locked memory (no re-pinning per volume)
            dt = time.perf_counter() - t0
            with torch.cuda.stream(self.copy_stream):
                dev = buf.to(dev_id, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            self._pin_ev[slot] = ev
        with self._lock:
            self.stats["stage_bytes"] += src.numel() * 4
            self.stats["stage_s"] += dt
        return dev, ev, img

    # ---- download side ----------------------------------------------------------------------------------------------------------------
    def _queue_d2h(self, res: VolumeResult, done: torch.cuda.Event) -> Tuple[int, torch.cuda.Event, tuple]:
        """DOWNLOAD WORKER (round 6; rounds 4-5: the launch thread): wait ON THE HOST for the volume's compute (`done`), then issue the D2H of its
        results on the download stream into a free pinned set.  Why not queue the copy behind `done` on the GPU as before: a stream that waits for an
        event keeps a barrier packet at the head of its hardware queue for the whole volume (~130 ms), HIP deals its streams onto a few IN-ORDER
        hardware queues, and whatever other stream shares that queue sits behind the barrier -- when it was the upload stream, the next volume's
        upload-done marker (and with it the next volume's first kernel) waited for the previous volume's whole D2H: 7.3 ms per volume, or 3, or 0,
        depending on how the streams had been dealt: the "bi-stable" 0.93-0.99 of the resident rate of round 5 (profiles/r06_cohort.md; with
        GPU_MAX_HW_QUEUES=1 always 7.3 ms, with 2 never).  Issued only once `done` has fired, the copy never parks a barrier in any queue."""
        done.synchronize()
        t0 = time.perf_counter()
        k = self._free_out.get()
        self.stats["launch_wait_s"] += time.perf_counter() - t0
        pins = self._pin_out[k]
        names = _RESULT_NAMES + (("overflow",) if res.overflow is not None else ())
        with torch.cuda.stream(self.down_stream):
            for name in names:
                t = getattr(res, name)
                key = (name, tuple(t.shape), t.dtype)
                if key not in pins:
                    pins[key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
                pins[key].copy_(t, non_blocking=True)             # (no record_stream: the download worker holds `res` until this copy has completed --
                #                                                    the allocator gets the blocks back by plain refcount, without per-allocation event polling)
            ev = torch.cuda.Event()
            ev.record(self.down_stream)
        return k, ev, tuple((name, tuple(getattr(res, name).shape), getattr(res, name).dtype) for name in names)

    def _collect(self, k: int, ev: torch.cuda.Event, keys: tuple, repeated: bool, held=None) -> Optional[VolumeResult]:
        """Download worker: wait for the D2H, copy the results out of the pinned set into memory the caller owns, free the set.  None = the
        fp16 range flag of the volume (4 bytes, rides along) is raised: the results must not be used.  ``held``: the device tensors being
        copied -- referenced until the copy is done."""
        try:
            ev.synchronize()
            held = None                                                # the device results may go back to the allocator now
            pins = self._pin_out[k]
            if len(keys) > 5 and int(pins[keys[5]][0]):
                return None
            t0 = time.perf_counter()
            if self.result_pool:
                dst = self._pool_sets[self._pool_next]                     # (only the download worker touches the ring)
                self._pool_next = (self._pool_next + 1) % self.result_pool
                for key in keys[:5]:
                    if key not in dst:
                        dst[key] = torch.empty(pins[key].shape, dtype=pins[key].dtype)
                outs = list(self._clone.map(lambda key: dst[key].copy_(pins[key]), keys[:5]))
            else:
                outs = list(self._clone.map(lambda key: torch.empty(pins[key].shape, dtype=pins[key].dtype).copy_(pins[key]), keys[:5]))
            dt = time.perf_counter() - t0
            with self._lock:
                self.stats["clone_bytes"] += sum(o.numel() * o.element_size() for o in outs)
                self.stats["clone_s"] += dt
            return VolumeResult(*outs, repeated_f32=repeated)
        finally:
            self._free_out.put(k)                                          # (the pinned buffers are reused by a later volume)

    def _download(self, res: VolumeResult, done: torch.cuda.Event) -> Optional[VolumeResult]:
        with torch.cuda.device(self.pipe.unet.device):
            k, ev, keys = self._queue_d2h(res, done)
            return self._collect(k, ev, keys, res.repeated_f32, res)

    def _download_async(self, res: VolumeResult, done: torch.cuda.Event) -> Future:
        return self._down.submit(self._download, res, done)

    def _finish(self, pending) -> VolumeResult:
        """Results of a queued volume whose download was started a volume ago; a volume whose fp16x3 segmentation left fp16's range is
        repeated in exact fp32 here (the check is lazy -- at download time -- so the overlap of the normal case is kept; never silent)."""
        i, res, done, dev, img, fut = pending
        cs = self.compute_stream if self.compute_stream is not None else torch.cuda.current_stream(self.pipe.unet.device)
        if self.keep_on_device:
            done.synchronize()
            if res.overflow is not None:
                raised = bool(int(res.overflow.item()))
                self._note_flag(raised, dev, cs)
                if raised:
                    with torch.cuda.stream(cs):
                        res = self.pipe.rerun_f32(dev, img)
                    cs.synchronize()
            for name in _RESULT_NAMES:                                    # produced on the runner's compute stream, handed to the caller's
                getattr(res, name).record_stream(torch.cuda.current_stream(self.pipe.unet.device))
            return res
        out = fut.result()
        if res.overflow is not None:
            self._note_flag(out is None, dev, cs)
        if out is None:
            with torch.cuda.stream(cs):
                res = self.pipe.rerun_f32(dev, img)
                done = torch.cuda.Event()
                done.record(cs)
            out = self._download_async(res, done).result()
        return out

    def _note_flag(self, raised: bool, dev: torch.Tensor, cs) -> None:
        """Streak bookkeeping of the fp16x3 range flag; when it drops a calibration FILE and the cohort has a board, the ranks agree on one
        recalibration now (this rank calibrates on the volume it holds, or mirrors the rank that did) instead of each on its next volume."""
        eng = self.pipe.unet
        if eng.note_volume_flag(raised) and self.board is not None:
            with torch.cuda.stream(cs):
                self.board.recalibrate(eng, lambda: eng.calibrate_volume(dev, self.pipe.tile_zyx, self.pipe.overlap_zyx, self.pipe.crop_zyx))

    def run(self, images: Sequence, rank: int = 0, world: int = 1, queue=None) -> Iterator[Tuple[int, VolumeResult]]:
        """Yield (index, result) for the volumes this worker processes, in its processing order.  ``queue`` (a
        ``parallel.VolumeQueue`` shared by all ranks) assigns volumes dynamically -- the worker claims one volume ahead, so that its
        upload overlaps the current compute; without a queue the static split index % world == rank is used.  Results are yielded
        ``LAG`` volumes behind their compute's queueing (their download and host copy run on the workers meanwhile)."""
        if queue is not None:
            order = iter(queue)
        else:
            order = iter([i for i in range(len(images)) if i % world == rank])
        cur = next(order, None)
        if cur is None:
            return
        # every image is fetched from ``images`` exactly ONCE (a lazy sequence -- dask_processing._Lazy -- reads and normalises the file
        # in __getitem__), on the upload worker, and travels with its upload: (device tensor, upload event, host image)
        nxt = self._up.submit(self._upload, lambda i=cur: images[i], 0)
        waiting: deque = deque()                                          # volumes whose compute is queued: (index, res, done, dev, img, download future | None)
        k = 0
        st, clock = self.stats, time.perf_counter
        caller = torch.cuda.current_stream(self.pipe.unet.device)
        cs = self.compute_stream if self.compute_stream is not None else caller
        cs.wait_stream(caller)                                            # whatever the caller queued before (engines, atlas) is visible to the volumes' kernels
        while cur is not None:
            t0 = clock()
            dev, ev, img = nxt.result()
            t1 = clock()
            if self.board is not None:
                self.board.poll(self.pipe.unet)                           # another rank's recalibration, if one was published since the last volume
            with torch.cuda.stream(cs):
                cs.wait_event(ev)
                dev.record_stream(cs)                                     # allocated on the copy stream, read by the compute stream
                res = self.pipe.run(dev, img, check=False)                # queued, not waited for; the range flag is read at download time
                done = torch.cuda.Event()
                done.record(cs)
            t2 = clock()
            following = next(order, None)                                 # claimed now: its host staging + H2D run behind this volume's compute
            if following is not None:
                nxt = self._up.submit(self._upload, lambda i=following: images[i], (k + 1) & 1)
            fut = None if self.keep_on_device else self._download_async(res, done)      # D2H queued behind `done`; the host copy on the worker
            t3 = clock()
            waiting.append((cur, res, done, dev, img, fut))
            st["t_upload_wait"] += t1 - t0; st["t_queue_compute"] += t2 - t1; st["t_issue_d2h"] += t3 - t2
            while len(waiting) > self.LAG:                                # hand out what is at least LAG volumes old
                p = waiting.popleft()
                t4 = clock()
                out = self._finish(p)
                st["t_result_wait"] += clock() - t4
                yield p[0], out
            cur, k = following, k + 1
        while waiting:
            p = waiting.popleft()
            yield p[0], self._finish(p)
