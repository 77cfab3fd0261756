"""One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Two ways the path shards; neither needs an all-reduce:

* ``replicas``  -- cohort throughput: volume v goes to rank v mod N, no data-path collective.
* ``tile shard``-- single-volume latency: rank 0's volume is broadcast (94 MB, ``broadcast_volume``); the 160
  independent tiles are split at tile granularity in the reference's z-major order (rank g gets a contiguous
  range balanced by per-tile work), every rank runs the U-Net on its tiles, and ONE all_gather of the kept
  centre blocks (2 x 94 MB fp32 per volume) gives every rank the full set before stitch / registration; the
  phi-resample of the two maps is sharded by atlas z-slab (``slab_range_for_rank`` + ``gather_slabs``).  Tiles
  are addressing into the replicated input volume, and activations have no halos because the reference
  zero-pads at tile borders, so there is no halo exchange.  The fp16 range flag is
  all_reduce(MAX)ed (``any_rank``) so that every rank repeats a volume in fp32 together.

The functions take the per-rank compute as a callable so that the collective logic is testable on CPU
with the gloo backend (tests/test_parallel_cpu.py).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def tile_range_for_rank(n_tiles: int, rank: int, world: int, costs: Optional[Sequence[float]] = None) -> Tuple[int, int]:
    """Contiguous split of the tile list.  Without ``costs``: balanced by count (the first n_tiles % world ranks take one extra
    tile).  With per-tile ``costs`` (``UNetEngine.tile_costs``: border tiles are cheaper, their kept centre is trimmed): the
    boundaries are placed where the running cost crosses k / world of the total, so that every rank gets the same WORK; every
    rank still gets a (possibly empty) contiguous range and the ranges tile [0, n_tiles) in rank order."""
    if costs is None:
        base, extra = divmod(n_tiles, world)
        begin = rank * base + min(rank, extra)
        return begin, begin + base + (1 if rank < extra else 0)
    if len(costs) != n_tiles:
        raise ValueError("one cost per tile")
    total = float(sum(costs))
    bounds, acc, t = [0], 0.0, 0
    for k in range(1, world):
        target = total * k / world
        while t < n_tiles and acc + 0.5 * float(costs[t]) <= target:       # a tile goes to the side its midpoint falls on
            acc += float(costs[t])
            t += 1
        bounds.append(t)
    bounds.append(n_tiles)
    return bounds[rank], bounds[rank + 1]


def volumes_for_rank(n_volumes: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_volumes, world))


def gather_blocks(local_blocks: torch.Tensor, n_tiles: int, group=None, costs: Optional[Sequence[float]] = None) -> torch.Tensor:
    """all_gather the per-rank centre blocks [n_local, C, ez, ey, ex] into [n_tiles, C, ez, ey, ex].

    Ranges are contiguous and ordered by rank, so concatenating the gathered pieces is the stitch order.
    Uneven ranges (n_tiles % world != 0) are padded to the largest range for the collective.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if not dist.is_initialized():
        return local_blocks
    rank = dist.get_rank(group)
    counts = [tile_range_for_rank(n_tiles, r, world, costs) for r in range(world)]
    max_n = max(e - b for b, e in counts)
    tail = local_blocks.shape[1:]
    if counts[rank][1] - counts[rank][0] != local_blocks.shape[0]:
        raise ValueError("local block count does not match this rank's tile range")
    send = local_blocks
    if send.shape[0] < max_n:
        pad = torch.zeros((max_n - send.shape[0], *tail), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], 0)
    send = send.contiguous()
    out = torch.empty((world * max_n, *tail), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if all(e - b == max_n for b, e in counts):
        return out
    pieces = [out[r * max_n: r * max_n + (e - b)] for r, (b, e) in enumerate(counts)]
    return torch.cat(pieces, 0)


def segment_tile_sharded(compute_blocks: Callable[[Tuple[int, int]], torch.Tensor], n_tiles: int, group=None,
                         costs: Optional[Sequence[float]] = None) -> torch.Tensor:
    """Every rank computes its tile range (balanced by ``costs`` when given), then all ranks hold all blocks."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    rng = tile_range_for_rank(n_tiles, rank, world, costs)
    return gather_blocks(compute_blocks(rng), n_tiles, group, costs)


# ---- input distribution, z-slab sharded resample, flag agreement ------------------------------------------------

def broadcast_volume(vol: Optional[torch.Tensor], shape: Sequence[int], device, src: int = 0, group=None) -> torch.Tensor:
    """The volume lives on rank ``src`` (the reference's Dask worker that loaded it, dask_processing.py:46-75); every other rank
    passes ``vol=None`` and receives it: one 94 MB broadcast (RCCL over xGMI ~0.6 ms), cheaper and simpler than per-rank
    8+8-slice halo sends because tiles are addressing into the whole volume."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        if vol is None:
            raise ValueError("broadcast_volume: no volume on a single rank")
        return vol
    rank = dist.get_rank(group)
    if rank == src:
        if vol is None or tuple(vol.shape) != tuple(shape):
            raise ValueError("broadcast_volume: the source rank must pass the volume with the announced shape")
        buf = vol.to(device=device, dtype=torch.float32).contiguous()
    else:
        buf = torch.empty(tuple(int(v) for v in shape), dtype=torch.float32, device=device)
    dist.broadcast(buf, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return buf


def slab_range_for_rank(n_slices: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous z-slab of the atlas grid for the sharded phi-resample (work per slice is uniform: balanced by count)."""
    return tile_range_for_rank(n_slices, rank, world)


def gather_slabs(local: torch.Tensor, n_slices: int, group=None) -> torch.Tensor:
    """all_gather z-slabs ``local`` [C, z_r, H, W] (rank r holds ``slab_range_for_rank(n_slices, r, world)``) into [C, n_slices, H, W].
    One collective; ragged slabs are padded to the largest."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ranges = [slab_range_for_rank(n_slices, r, world) for r in range(world)]
    if local.shape[1] != ranges[rank][1] - ranges[rank][0]:
        raise ValueError("local slab does not match this rank's z range")
    max_n = max(e - b for b, e in ranges)
    Cn, _, H, W = local.shape
    send = local.transpose(0, 1).contiguous()                      # [z, C, H, W]: slices are the gathered unit
    if send.shape[0] < max_n:
        send = torch.cat([send, torch.zeros((max_n - send.shape[0], Cn, H, W), dtype=send.dtype, device=send.device)], 0)
    out = torch.empty((world * max_n, Cn, H, W), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if not all(e - b == max_n for b, e in ranges):
        out = torch.cat([out[r * max_n: r * max_n + (e - b)] for r, (b, e) in enumerate(ranges)], 0)
    return out.transpose(0, 1).contiguous()


def any_rank(flag: torch.Tensor, group=None) -> torch.Tensor:
    """all_reduce(MAX) of a small integer tensor (the fp16 range flag): after it every rank holds the same verdict."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return flag


# ---- cohort scheduling: a volume queue instead of a static assignment -------------------------------------

class VolumeQueue:
    """Dynamic assignment of the volumes of a cohort to persistent per-GPU workers: every rank claims the next unprocessed index
    with one atomic add on the process group's key-value store (TCPStore on rank 0).  A rank that is slower -- its volumes needed
    the fp32 repeat, its GPU is clocked lower (devices differ by up to 12 %, MI355X_MICROARCH.md) -- simply claims fewer volumes,
    where the static ``v % world`` split of ``volumes_for_rank`` (and the reference's Dask graph, which rebuilds the models per
    task, dask_processing.py:77,170) would wait for it.  Without an initialised process group it is a local counter."""

    _serial = 0

    def __init__(self, n_volumes: int, name: Optional[str] = None, store=None):
        """``store``: a torch.distributed Store shared by the ranks (default: the default process group's); collective in the sense that
        every rank must construct its queues in the same order (or pass the same ``name``).  The size travels with the key: the first
        rank to arrive publishes ``n``, every other rank checks that it was about to iterate over the same cohort."""
        self.n = int(n_volumes)
        self._local = 0
        self._store = None
        self._done = False
        self.claimed: List[int] = []                # what THIS rank took (a driver can diff it against the finished results)
        if store is not None or (dist.is_initialized() and dist.get_world_size() > 1):
            if store is None:
                from torch.distributed import distributed_c10d
                store = distributed_c10d._get_default_store()      # (private API: the one store every rank already shares)
            self._store = store
            if name is None:                       # every rank constructs its queues in the same order: same key on all ranks
                name = f"oai_volume_queue_{VolumeQueue._serial}"
                VolumeQueue._serial += 1
            self._key = name
            if int(self._store.add(self._key + "_ranks", 1)) == 1:
                self._store.set(self._key + "_n", str(self.n))
            else:
                self._store.wait([self._key + "_n"])
                n_pub = int(self._store.get(self._key + "_n"))
                if n_pub != self.n:
                    raise RuntimeError(f"VolumeQueue '{name}': this rank iterates over {self.n} volumes, the rank that created the key over "
                                       f"{n_pub} (a queue constructed on some ranks only, or a reused name)")

    def claim(self) -> Optional[int]:
        """Index of the next volume, or None when the cohort is exhausted.  Each index is handed out exactly once across ranks."""
        if self._done:
            return None
        if self._store is None:
            i = self._local
            self._local += 1
        else:
            i = int(self._store.add(self._key, 1)) - 1
        if i < self.n:
            self.claimed.append(i)
            return i
        self._done = True                          # (the counter key stays: a late rank must still see an exhausted queue, not a fresh one)
        return None

    def __iter__(self):
        while True:
            i = self.claim()
            if i is None:
                return
            yield i


# ---- one fp16x3 calibration per checkpoint and cohort, whatever rank sees which volume first -----------------

_cal_serial = 0


def sync_calibration(engine, calibrate_fn: Callable[[], None], store=None, name: Optional[str] = None, rank: Optional[int] = None,
                     world: Optional[int] = None) -> List[int]:
    """Every rank leaves with RANK 0's activation exponents: rank 0 calibrates (``calibrate_fn()``, unless its engine already holds a
    calibration -- e.g. from the checkpoint's sidecar file) and publishes the 18 exponents on the process group's key-value store;
    the other ranks wait for the key and ``set_act_exponents``.  Without this every rank of ``process_cohort`` would calibrate on
    the first volume the queue happens to hand it, and a volume's last bits would depend on which rank claimed it.  No process
    group: just ``calibrate_fn()`` when needed.  ``engine`` needs ``act_exponents()``, ``set_act_exponents()`` (UNetEngine)."""
    global _cal_serial
    import json
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    exps, cal = engine.act_exponents()
    if world == 1 and store is None:
        if not cal:
            calibrate_fn()
        return engine.act_exponents()[0]
    if store is None:
        from torch.distributed import distributed_c10d
        store = distributed_c10d._get_default_store()
    if name is None:                                   # every rank calls in the same order: same key everywhere
        name = f"oai_fp16cal_{_cal_serial}"
        _cal_serial += 1
    if rank == 0:
        if not cal:
            calibrate_fn()
        exps, cal = engine.act_exponents()
        store.set(name, json.dumps({"act_exponents": exps, "calibrated": bool(cal),
                                    "weights_sha256": getattr(engine, "weights_sha256", None)}))
        return exps
    store.wait([name])
    doc = json.loads(store.get(name))
    if doc.get("weights_sha256") != getattr(engine, "weights_sha256", None):
        raise RuntimeError("sync_calibration: rank 0 holds other weights than this rank")
    if doc["calibrated"]:
        engine.set_act_exponents(doc["act_exponents"])
    return doc["act_exponents"]
