"""CPU, world_size 2, gloo: the tile-shard collective logic reproduces the single-process result.
The per-rank compute is the oracle (tests may use it); the sharding / gather code is the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oai_analysis_2_amd import parallel
from oai_analysis_2_amd.synth import make_unet_state_dict, make_volume


def test_tile_ranges_cover_exactly():
    for n in (160, 75, 7, 1):
        for w in (1, 2, 3, 4, 8):
            rs = [parallel.tile_range_for_rank(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in rs]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.tile_range_for_rank(160, 3, 8) == (60, 80)       # 20 tiles per GPU
    assert parallel.volumes_for_rank(10, 1, 4) == [1, 5, 9]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_tiles, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import seg as oseg
    vol = make_volume(9, (12, 40, 40))
    sd = make_unet_state_dict(seed=2, width_div=4)
    patch, ovl = (16, 16, 8), (4, 4, 2)          # x,y,z
    tiles, g = oseg.partition(vol, patch, ovl)
    assert g["n_tiles"] == n_tiles

    def compute(rng):
        b, e = rng
        logits = oseg.unet_forward(torch.from_numpy(tiles[b:e]), sd) if e > b else torch.zeros((0, 2, 8, 16, 16))
        o = g["overlap"]
        t = g["tile"]
        return torch.sigmoid(logits)[:, :, o[0]:t[0] - o[0], o[1]:t[1] - o[1], o[2]:t[2] - o[2]].contiguous()

    blocks = parallel.segment_tile_sharded(compute, n_tiles)
    np.save(os.path.join(out_dir, f"blocks_{rank}.npy"), blocks.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_tile_shard_gather_matches_single_process(tmp_path, world):
    from oracle import seg as oseg
    vol = make_volume(9, (12, 40, 40))
    sd = make_unet_state_dict(seed=2, width_div=4)
    tiles, g = oseg.partition(vol, (16, 16, 8), (4, 4, 2))
    n_tiles = g["n_tiles"]
    assert n_tiles % world != 0                         # both exercise the ragged, padded gather
    o, t = g["overlap"], g["tile"]
    ref = torch.sigmoid(oseg.unet_forward(torch.from_numpy(tiles), sd))[:, :, o[0]:t[0] - o[0], o[1]:t[1] - o[1], o[2]:t[2] - o[2]].numpy()
    mp.spawn(_worker, args=(world, _free_port(), n_tiles, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"blocks_{r}.npy")
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6)
    # and stitching the gathered blocks equals the oracle's assemble
    fc = oseg.assemble(np.pad(got[:, 0], ((0, 0), (o[0], o[0]), (o[1], o[1]), (o[2], o[2]))), g, crop_size_xyz=(4, 4, 2))
    fc_ref, _ = oseg.segment(vol, sd, (16, 16, 8), (4, 4, 2))
    np.testing.assert_allclose(fc, fc_ref, rtol=0, atol=1e-6)
