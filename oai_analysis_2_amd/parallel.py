"""One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

Two ways the path shards; neither needs an all-reduce:

* ``replicas``  -- cohort throughput: volume v goes to rank v mod N, no data-path collective.
* ``tile shard``-- single-volume latency: the 160 independent tiles are split at tile granularity in the
  reference's z-major order (rank g gets a contiguous range, <= 1 tile imbalance), every rank runs the
  U-Net on its tiles, and ONE all_gather of the kept centre blocks (2 x 94 MB fp32 per volume) gives every
  rank the full set before stitch / registration / resample.  Input "halo" needs no exchange: tiles are
  addressing into the replicated input volume (94 MB broadcast), and activations have no halos because the
  reference zero-pads at tile borders.

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
