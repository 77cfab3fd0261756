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


# (name, kind, cin, cout, output level) of the reference UNet (networks.py:43-64); kind: 0 / 1 = k3 conv / transposed conv, 2 = k2s2 up-conv, 3 = 1x1x1 head
_UNET_LAYERS = (("ec0", 0, 1, 32, 0), ("ec1", 0, 32, 64, 0), ("ec2", 0, 64, 64, 1), ("ec3", 0, 64, 128, 1), ("ec4", 0, 128, 128, 2),
                ("ec5", 0, 128, 256, 2), ("ec6", 0, 256, 256, 3), ("ec7", 0, 256, 512, 3), ("dc9", 2, 512, 512, 2), ("dc8", 1, 768, 256, 2),
                ("dc7", 1, 256, 256, 2), ("dc6", 2, 256, 256, 1), ("dc5", 1, 384, 128, 1), ("dc4", 1, 128, 128, 1), ("dc3", 2, 128, 128, 0),
                ("dc2", 1, 192, 64, 0), ("dc1", 1, 64, 64, 0), ("dc0", 3, 64, 2, 0))


def tile_costs_host(size_zyx: Sequence[int], tile_zyx: Sequence[int], overlap_zyx: Sequence[int], crop_zyx: Optional[Sequence[int]] = None,
                    layers=_UNET_LAYERS) -> List[float]:
    """FLOPs of every tile as ``oai_segment_tiles`` computes it -- the host mirror of ``oai_unet_tile_costs`` (csrc/unet.hip: keep_interval,
    plan_regions, layer_flops) for planning a tile shard WITHOUT a device handle (a scheduler, bench.py --dry-run, the CPU tests; on a GPU box
    ``UNetEngine.tile_costs`` is the same list, asserted in tests/test_fullsize_gpu.py).  Border tiles are cheaper: Partition.assemble zeroes a
    frame of ``crop`` voxels and trims to the image (image_transforms.py:504-513), and every layer computes only the box its consumers need."""
    size, tile, ovl = [int(v) for v in size_zyx], [int(v) for v in tile_zyx], [int(v) for v in overlap_zyx]
    crop = [int(v) for v in crop_zyx] if crop_zyx is not None else [0, 0, 0]
    eff = [t - 2 * o for t, o in zip(tile, ovl)]
    grid = [-(-s // e) for s, e in zip(size, eff)]
    dims = [[t >> l for t in tile] for l in range(4)]

    def grow(b, lvl):
        return ([max(0, a - 1) for a in b[0]], [min(f, h + 1) for h, f in zip(b[1], dims[lvl])])

    def halve(b):
        return ([a // 2 for a in b[0]], [(h + 1) // 2 for h in b[1]])

    costs = []
    for t in range(grid[0] * grid[1] * grid[2]):
        idx = (t // (grid[1] * grid[2]), (t // grid[2]) % grid[1], t % grid[2])
        lo = [ovl[i] + max(0, crop[i] - idx[i] * eff[i]) for i in range(3)]
        hi = [ovl[i] + min(eff[i], size[i] - crop[i] - idx[i] * eff[i]) for i in range(3)]
        if any(h <= l for l, h in zip(lo, hi)):
            costs.append(0.0)
            continue
        need = {name: ([0, 0, 0], list(dims[lvl])) for name, _, _, _, lvl in layers}
        need["dc0"] = need["dc1"] = (lo, hi)
        need["dc2"] = grow(need["dc1"], 0)
        need["dc3"] = grow(need["dc2"], 0)
        need["dc4"] = halve(need["dc3"])
        need["dc5"] = grow(need["dc4"], 1)
        need["dc6"] = grow(need["dc5"], 1)
        need["dc7"] = halve(need["dc6"])
        need["dc8"] = grow(need["dc7"], 2)
        f = 0.0
        for name, kind, cin, cout, _ in layers:
            b = need[name]
            vox = float((b[1][0] - b[0][0]) * (b[1][1] - b[0][1]) * (b[1][2] - b[0][2]))
            f += 2.0 * vox * (27 if kind in (0, 1) else 1) * cin * cout
        costs.append(f)
    return costs


def volumes_for_rank(n_volumes: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_volumes, world))


class GatheredBlocks:
    """What one all_gather of per-rank tile ranges leaves behind: ``buffer`` [world * stride, C, ez, ey, ex], rank r's blocks in slots
    [r * stride, r * stride + (bounds[r + 1] - bounds[r])).  ``UNetEngine.stitch`` reads it through ``oai_stitch_blocks_ranged`` -- no
    compacting copy; ``compact()`` is the plain [n_tiles, ...] tensor for callers that want one (a view when the ranges are equal)."""

    def __init__(self, buffer: torch.Tensor, bounds: Sequence[int], stride: int):
        self.buffer, self.bounds, self.stride = buffer, [int(v) for v in bounds], int(stride)

    @property
    def n_tiles(self) -> int:
        return self.bounds[-1]

    def slot(self, rank: int) -> torch.Tensor:
        """The view of ``buffer`` that rank ``rank`` fills: [count_r, C, ez, ey, ex]."""
        return self.buffer[rank * self.stride: rank * self.stride + self.bounds[rank + 1] - self.bounds[rank]]

    def compact(self) -> torch.Tensor:
        world = len(self.bounds) - 1
        if all(self.bounds[r + 1] - self.bounds[r] == self.stride for r in range(world)):
            return self.buffer
        return torch.cat([self.slot(r) for r in range(world)], 0)


def _tile_bounds(n_tiles: int, world: int, costs) -> List[int]:
    ranges = [tile_range_for_rank(n_tiles, r, world, costs) for r in range(world)]
    return [0] + [e for _, e in ranges]


def gather_blocks_padded(local_blocks: torch.Tensor, n_tiles: int, group=None, costs: Optional[Sequence[float]] = None,
                         into: Optional[GatheredBlocks] = None) -> GatheredBlocks:
    """ONE all_gather of the per-rank centre blocks, without a pad + cat on the send side or a compacting cat behind it (VERDICT r4
    #4b: 2 x 189 MB of copies per volume around a 189 MB collective): the gather buffer holds ``stride`` = the largest range's
    blocks per rank, every rank's blocks go (or already are: ``into`` from ``alloc_gather``, ``local_blocks`` = its ``slot(rank)``)
    in their slot, the collective runs IN PLACE on that buffer, and the stitch reads the ragged layout through a table."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    bounds = _tile_bounds(n_tiles, world, costs)
    stride = max(bounds[r + 1] - bounds[r] for r in range(world))
    if bounds[rank + 1] - bounds[rank] != local_blocks.shape[0]:
        raise ValueError("local block count does not match this rank's tile range")
    g = into
    if g is None:
        g = GatheredBlocks(torch.empty((world * stride, *local_blocks.shape[1:]), dtype=local_blocks.dtype, device=local_blocks.device), bounds, stride)
    elif g.bounds != bounds or g.stride != stride or g.buffer.shape[1:] != local_blocks.shape[1:]:
        raise ValueError("gather buffer was allocated for another split")
    mine = g.slot(rank)
    if local_blocks.data_ptr() != mine.data_ptr() and local_blocks.shape[0]:
        mine.copy_(local_blocks)                                   # (callers that computed straight into slot(rank) skip this)
    if world > 1:
        send = g.buffer[rank * stride: (rank + 1) * stride]          # the whole slot: its unused tail travels as padding
        dist.all_gather_into_tensor(g.buffer, send, group=group)
    return g


def alloc_gather(n_tiles: int, block_shape: Sequence[int], dtype, device, group=None, costs: Optional[Sequence[float]] = None) -> GatheredBlocks:
    """The gather buffer of ``gather_blocks_padded`` ahead of the compute, so that a rank's kernels write their blocks straight into
    ``slot(rank)`` (``UNetEngine.segment_tiles(out=...)``)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    bounds = _tile_bounds(n_tiles, world, costs)
    stride = max(bounds[r + 1] - bounds[r] for r in range(world))
    return GatheredBlocks(torch.empty((world * stride, *[int(v) for v in block_shape]), dtype=dtype, device=device), bounds, stride)


def gather_blocks(local_blocks: torch.Tensor, n_tiles: int, group=None, costs: Optional[Sequence[float]] = None) -> torch.Tensor:
    """all_gather the per-rank centre blocks [n_local, C, ez, ey, ex] into ONE tensor [n_tiles, C, ez, ey, ex] in stitch order
    (ranges are contiguous and ordered by rank).  Convenience form of ``gather_blocks_padded`` for callers that want the plain
    tensor: ragged ranges cost one compacting copy here, none through ``GatheredBlocks`` + ``UNetEngine.stitch``."""
    if not dist.is_initialized():
        return local_blocks
    return gather_blocks_padded(local_blocks, n_tiles, group, costs).compact()


def segment_tile_sharded(compute_blocks: Callable[..., torch.Tensor], n_tiles: int, group=None,
                         costs: Optional[Sequence[float]] = None, block_shape: Optional[Sequence[int]] = None, dtype=torch.float32,
                         device=None):
    """Every rank computes its tile range (balanced by ``costs`` when given), then all ranks hold all blocks.

    ``block_shape`` None: ``compute_blocks(rng)`` returns the rank's blocks, the result is the plain [n_tiles, ...] tensor.
    ``block_shape`` = (C, ez, ey, ex): the gather buffer is allocated first, ``compute_blocks(rng, out)`` writes into this rank's slot
    of it, the collective runs in place and the result is a ``GatheredBlocks`` (no copy on either side of the collective)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    rng = tile_range_for_rank(n_tiles, rank, world, costs)
    if block_shape is None:
        return gather_blocks(compute_blocks(rng), n_tiles, group, costs)
    g = alloc_gather(n_tiles, block_shape, dtype, device, group, costs)
    mine = g.slot(rank)
    got = compute_blocks(rng, mine)
    return gather_blocks_padded(mine if got is None else got, n_tiles, group, costs, into=g)


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
    One collective PER MAP, each gathering contiguous [z_r, H, W] slabs straight from the resample kernel's [C][z][y][x] output into the
    result's [C][z][y][x] -- no transpose on either side (the one-collective form moved 2 x 189 MB through two transposes around a
    189 MB gather; VERDICT r4 #4b).  Equal slabs (160 atlas slices over 8 ranks): the collective writes the result in place.  Ragged
    slabs are gathered into ``max`` slices per rank and compacted with one copy per map."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ranges = [slab_range_for_rank(n_slices, r, world) for r in range(world)]
    if local.shape[1] != ranges[rank][1] - ranges[rank][0]:
        raise ValueError("local slab does not match this rank's z range")
    max_n = max(e - b for b, e in ranges)
    Cn, _, H, W = local.shape
    equal = all(e - b == max_n for b, e in ranges)
    out = torch.empty((Cn, n_slices, H, W), dtype=local.dtype, device=local.device)
    local = local.contiguous()
    for c in range(Cn):
        if equal:
            dist.all_gather_into_tensor(out[c], local[c], group=group)
            continue
        send = local[c]
        if send.shape[0] < max_n:
            buf = torch.zeros((max_n, H, W), dtype=local.dtype, device=local.device)
            buf[:send.shape[0]] = send
            send = buf
        padded = torch.empty((world * max_n, H, W), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(padded, send, group=group)
        for r, (b, e) in enumerate(ranges):
            out[c, b:e] = padded[r * max_n: r * max_n + (e - b)]
    return out


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
    """Every rank leaves with RANK 0's calibration OUTCOME -- not only its exponents: rank 0 calibrates (``calibrate_fn()``,
    unless its engine already holds a verdict -- e.g. exponents from the checkpoint's sidecar file) and publishes ONE status on the process
    group's key-value store; the other ranks wait for the key and mirror it:

        "calibrated"   exponents set on every rank (``set_act_exponents``)
        "refused_f32"  the calibration did not settle on rank 0: EVERY rank runs exact fp32 (``refuse_fp16``) -- otherwise rank 0 would run
                       f32 and the others fp16x3 with exponents of their own first volume: the rank-dependence this function removes
        "no_census"    the network records no range census (narrow test networks): exponents stay 0 everywhere, nobody calibrates again
        "error"        ``calibrate_fn`` raised on rank 0: the message is published (a key is ALWAYS published, so nobody sits in
                       ``store.wait`` until the store's timeout) and every rank raises

    Without this every rank of ``process_cohort`` would calibrate on the first volume the queue happens to hand it, and a volume's last bits
    would depend on which rank claimed it.  No process group: just ``calibrate_fn()`` when needed.  ``engine`` needs ``act_exponents()``,
    ``set_act_exponents()`` and -- UNetEngine has them; optional on stand-ins -- ``calibration_status()``, ``refuse_fp16()``,
    ``mark_no_census()``, ``effective_precision``."""
    global _cal_serial
    import json
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0

    def status_of(eng) -> str:                          # one source of "calibrated?": the engine's own verdict when it has one
        if hasattr(eng, "calibration_status"):
            return eng.calibration_status()
        return "calibrated" if eng.act_exponents()[1] else "uncalibrated"

    if world == 1 and store is None:
        if status_of(engine) == "uncalibrated":
            calibrate_fn()
        return engine.act_exponents()[0]
    if store is None:
        from torch.distributed import distributed_c10d
        store = distributed_c10d._get_default_store()
    if name is None:                                   # every rank calls in the same order: same key everywhere
        name = f"oai_fp16cal_{_cal_serial}"
        _cal_serial += 1
    if rank == 0:
        doc = {"weights_sha256": getattr(engine, "weights_sha256", None)}
        try:
            if status_of(engine) == "uncalibrated":
                calibrate_fn()
            st = status_of(engine)
            if st == "uncalibrated":                    # calibrate_fn returned without a verdict: treat as an error, never publish "go on as you are"
                raise RuntimeError("calibrate_fn left the engine uncalibrated")
            doc.update(status=st, act_exponents=engine.act_exponents()[0], precision=getattr(engine, "effective_precision", None))
        except Exception as exc:                        # noqa: BLE001 - whatever went wrong, the other ranks must hear about it
            doc.update(status="error", message=f"{type(exc).__name__}: {exc}")
            store.set(name, json.dumps(doc))
            raise
        store.set(name, json.dumps(doc))
        return doc["act_exponents"]
    store.wait([name])
    doc = json.loads(store.get(name))
    try:
        _mirror_calibration_doc(engine, doc)
    except RuntimeError as exc:
        raise RuntimeError(f"sync_calibration: {exc} (rank 0)") from None
    return doc["act_exponents"]


def _mirror_calibration_doc(engine, doc) -> None:
    """Take over a published calibration outcome (the receiving half of ``sync_calibration``)."""
    if doc.get("status") == "error":
        raise RuntimeError(f"calibration failed on the publishing rank ({doc.get('message')})")
    if doc.get("weights_sha256") != getattr(engine, "weights_sha256", None):
        raise RuntimeError("the publishing rank holds other weights than this rank")
    st = doc["status"]
    if st == "calibrated":
        engine.set_act_exponents(doc["act_exponents"])
    elif st == "refused_f32":
        if not hasattr(engine, "refuse_fp16"):
            raise RuntimeError("the publishing rank refused fp16x3 and this engine cannot mirror it")
        engine.refuse_fp16("mirrored from the publishing rank")
    elif st == "no_census":
        if hasattr(engine, "mark_no_census"):
            engine.mark_no_census()
    else:
        raise RuntimeError(f"unknown calibration status {st!r}")


class CalibrationBoard:
    """ONE recalibration for all ranks of a volume-parallel cohort when a calibration FILE turns out not to fit the data.

    ``UNetEngine.note_volume_flag`` drops a sidecar's calibration after three flagged volumes in a row.  In ``process_cohort`` every rank
    keeps its own streak and pulls volumes from a shared queue asynchronously: left alone, each rank would recalibrate on whichever volume
    it sees next, and a volume's last bits would depend on the rank again -- what ``sync_calibration`` exists to prevent.  The ranks share no
    lockstep point (no collective per volume), so the agreement goes through the process group's key-value store:

    * the first rank whose engine drops the calibration CLAIMS the epoch (``store.add`` on a counter: exactly one caller sees 1),
      recalibrates on the volume it holds and publishes the outcome (status + exponents, like ``sync_calibration``);
    * a rank that drops later finds the epoch claimed, waits for the publication and mirrors it instead of calibrating;
    * every rank ``poll``s (non-blocking ``store.check``) before it queues a volume and mirrors a publication it has not seen yet.

    Volumes already queued under the old exponents finish under them (they are flagged and repeated in exact fp32 if they do not fit --
    never silently wrong); from the first poll after the publication on, all ranks run the same exponents.  A key is ALWAYS published
    by the claimant (status "error" when its calibration raised), so a waiting rank cannot hang until the store's timeout."""

    def __init__(self, store=None, name: str = "oai_fp16recal"):
        if store is None and dist.is_initialized():
            from torch.distributed import distributed_c10d
            store = distributed_c10d._get_default_store()
        self.store, self.name, self.epoch = store, name, 0

    def _key(self, what: str) -> str:
        return f"{self.name}/{what}{self.epoch + 1}"

    def poll(self, engine) -> bool:
        """Mirror every publication this rank has not seen yet; never blocks.  True if the engine's calibration changed."""
        changed = False
        while self.store is not None and self.store.check([self._key("e")]):
            import json
            _mirror_calibration_doc(engine, json.loads(self.store.get(self._key("e"))))
            self.epoch += 1
            changed = True
        return changed

    def recalibrate(self, engine, calibrate_fn: Callable[[], None]) -> bool:
        """Called by a rank whose engine has just dropped its calibration.  True if THIS rank calibrated (``calibrate_fn()``), False if it
        mirrored another rank's outcome."""
        import json
        if self.store is None:
            calibrate_fn()
            return True
        if self.poll(engine):                       # somebody published since this rank last looked
            return False
        if self.store.add(self._key("claim"), 1) != 1:
            self.store.wait([self._key("e")])
            self.poll(engine)
            return False
        doc = {"weights_sha256": getattr(engine, "weights_sha256", None)}
        try:
            calibrate_fn()
            st = engine.calibration_status() if hasattr(engine, "calibration_status") else ("calibrated" if engine.act_exponents()[1] else "uncalibrated")
            if st == "uncalibrated":
                raise RuntimeError("calibrate_fn left the engine uncalibrated")
            doc.update(status=st, act_exponents=engine.act_exponents()[0], precision=getattr(engine, "effective_precision", None))
        except Exception as exc:                    # noqa: BLE001 - the waiting ranks must hear about it
            doc.update(status="error", message=f"{type(exc).__name__}: {exc}")
            self.store.set(self._key("e"), json.dumps(doc))
            self.epoch += 1
            raise
        self.store.set(self._key("e"), json.dumps(doc))
        self.epoch += 1
        return True

