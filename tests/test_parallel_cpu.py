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


def test_cost_weighted_tile_ranges():
    """ranges stay contiguous and ordered; each rank's work is within one tile of the ideal share; skewed costs move the cuts"""
    rng = np.random.default_rng(0)
    for n, w in ((160, 8), (75, 4), (7, 3), (5, 8)):
        costs = rng.uniform(0.5, 2.0, n).tolist()
        rs = [parallel.tile_range_for_rank(n, r, w, costs) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
        share = sum(costs) / w
        for b, e in rs:
            assert abs(sum(costs[b:e]) - share) <= max(costs) + 1e-9
    # the BASELINE geometry: z-rows 0 and 9 are cheap border rows -> the edge ranks take more tiles than 20
    costs = [0.6 if i // 16 in (0, 9) else 1.0 for i in range(160)]
    rs = [parallel.tile_range_for_rank(160, r, 8, costs) for r in range(8)]
    assert rs[0][1] - rs[0][0] > 20 and rs[7][1] - rs[7][0] > 20 and rs[3][1] - rs[3][0] < 20
    assert parallel.tile_range_for_rank(160, 3, 8, [1.0] * 160) == (60, 80)        # uniform costs = the count split
    with pytest.raises(ValueError):
        parallel.tile_range_for_rank(10, 0, 2, [1.0] * 9)


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

    costs = [1.0 + (i % 5) for i in range(n_tiles)] if os.environ.get("OAI_TEST_COSTS") == "1" else None
    blocks = parallel.segment_tile_sharded(compute, n_tiles, None, costs)
    np.save(os.path.join(out_dir, f"blocks_{rank}.npy"), blocks.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,weighted", [(2, False), (4, False), (3, True)])
def test_tile_shard_gather_matches_single_process(tmp_path, world, weighted, monkeypatch):
    from oracle import seg as oseg
    vol = make_volume(9, (12, 40, 40))
    sd = make_unet_state_dict(seed=2, width_div=4)
    tiles, g = oseg.partition(vol, (16, 16, 8), (4, 4, 2))
    n_tiles = g["n_tiles"]
    assert weighted or n_tiles % world != 0             # ragged ranges (by count, or by cost) -> the padded gather
    monkeypatch.setenv("OAI_TEST_COSTS", "1" if weighted else "0")          # (inherited by the spawned ranks)
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


def _worker_dist(rank, world, port, out_dir):
    """broadcast_volume / any_rank / gather_slabs of the tile-shard mode on CPU tensors over gloo."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shape, nz = (6, 10, 12), 11
    src = torch.arange(6 * 10 * 12, dtype=torch.float32).reshape(shape) * 0.5
    vol = parallel.broadcast_volume(src if rank == 0 else None, shape, "cpu", 0)
    ok = torch.equal(vol, src)
    try:                                                        # the source rank must hold the announced shape
        if rank == 0:
            parallel.broadcast_volume(None, shape, "cpu", 0)
            ok = False
    except ValueError:
        pass
    flag = parallel.any_rank(torch.tensor([1 if rank == world - 1 else 0], dtype=torch.int32))
    ok = ok and int(flag) == 1
    ok = ok and int(parallel.any_rank(torch.zeros(1, dtype=torch.int32))) == 0
    b, e = parallel.slab_range_for_rank(nz, rank, world)
    full = torch.arange(2 * nz * 3 * 4, dtype=torch.float32).reshape(2, nz, 3, 4)
    got = parallel.gather_slabs(full[:, b:e].contiguous(), nz)
    ok = ok and torch.equal(got, full)
    try:
        parallel.gather_slabs(full[:, b:e + 1].contiguous() if e < nz else full[:, b - 1:e].contiguous(), nz)
        ok = False
    except ValueError:
        pass
    with open(os.path.join(out_dir, f"ok_{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_broadcast_flag_agreement_and_slab_gather(tmp_path, world):
    assert [parallel.slab_range_for_rank(160, r, 8) for r in (0, 7)] == [(0, 20), (140, 160)]
    mp.spawn(_worker_dist, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok_{r}").read_text() == "1" for r in range(world))


@pytest.mark.parametrize("mode,world", [("tileshard", 2), ("tileshard", 8), ("replicas", 8), ("cohort", 8)])
def test_bench_spawns_its_own_ranks(mode, world):
    """`python bench.py --gpus N` with no launcher around it must start its ranks itself: exercised with --dry-run
    (CPU tensors over gloo, the product's parallel.py, no kernels) -- at the world size of the driver's 8-GPU run in all three modes
    (VERDICT r4 #4a: these passed by hand only), tileshard with the BASELINE geometry's 160-tile cost vector (ragged 23 / 19-tile ranges
    through the in-place gather and the stitch's slot table) and its 160 atlas slices."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--dry-run", "--mode", mode, "--steps", "2"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["world_size"] == world and out["backend"] == "gloo" and out["dry_run"] and out["mode"] == mode and out["value"] is None
    if mode == "tileshard" and world == 8:
        assert [e - b for b, e in out["tile_ranges"]] == [23, 19, 19, 19, 19, 19, 19, 23]          # the split of profiles/r04_tileshard_projection.md
    if mode == "cohort":
        # VERDICT r5 #4c: the cohort line carries the PER-RANK streamed-from-host leg next to the resident rate -- here the gather + aggregation of
        # the ranks' rows (steady ms, volumes/s, host GB/s) over gloo, with stand-in rows 130 + rank ms
        sf = out["streamed_from_host"]
        assert sf["ranks"] == world and sf["steady_ms_per_volume"] == {"min": 130.0, "max": 130.0 + world - 1}
        assert abs(sf["aggregate_steady_volumes_per_s"] - sum(1e3 / (130.0 + r) for r in range(world))) < 1e-9
        assert abs(sf["vs_resident"] - 129.0 / (130.0 + world - 1)) < 1e-12
    if world == 2:
        # without GPUs a real multi-GPU run must exit non-zero cleanly, before touching a device
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
        if not torch.cuda.is_available():
            assert r.returncode == 2 and "requested" in r.stderr


def _worker8(rank, world, port, out_dir):
    """World 8 over gloo with the REAL split of the BASELINE volume: 160 tiles under the trimmed per-tile costs, 160 atlas slices."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    n_tiles, nz = 160, 160
    costs = parallel.tile_costs_host((160, 384, 384), (32, 128, 128), (8, 16, 16), (8, 16, 16))
    shape = (2, 2, 3, 4)
    want = torch.arange(n_tiles, dtype=torch.float32)[:, None, None, None, None] + 0.25 * torch.arange(2, dtype=torch.float32)[None, :, None, None, None] + torch.zeros((n_tiles, *shape))
    seen = {}

    def compute(rng, out):                                  # writes this rank's blocks straight into its slot of the gather buffer
        seen["rng"] = rng
        out.copy_(want[rng[0]:rng[1]])

    g = parallel.segment_tile_sharded(compute, n_tiles, None, costs, block_shape=shape, dtype=torch.float32, device="cpu")
    ok = g.n_tiles == n_tiles and g.stride == 23 and seen["rng"] == tuple(parallel.tile_range_for_rank(n_tiles, rank, world, costs))
    ok = ok and [g.bounds[r + 1] - g.bounds[r] for r in range(world)] == [23, 19, 19, 19, 19, 19, 19, 23]
    for t in range(n_tiles):                                # what oai_stitch_blocks_ranged does: tile -> (range, slot) through the bounds table
        r = max(i for i in range(world) if g.bounds[i] <= t)
        ok = ok and torch.equal(g.buffer[r * g.stride + t - g.bounds[r]], want[t])
    ok = ok and torch.equal(g.compact(), want)
    # the plain-tensor form (callable without `out`) gives the same list
    ok = ok and torch.equal(parallel.segment_tile_sharded(lambda rng: want[rng[0]:rng[1]].clone(), n_tiles, None, costs), want)
    # 160 atlas slices in 8 equal slabs: gathered per map straight into [C, z, y, x]
    full = torch.arange(2 * nz * 5 * 6, dtype=torch.float32).reshape(2, nz, 5, 6)
    b, e = parallel.slab_range_for_rank(nz, rank, world)
    ok = ok and (e - b) == 20 and torch.equal(parallel.gather_slabs(full[:, b:e].contiguous(), nz), full)
    # ... and a ragged slab count
    full2 = torch.arange(2 * 75 * 2 * 3, dtype=torch.float32).reshape(2, 75, 2, 3)
    b, e = parallel.slab_range_for_rank(75, rank, world)
    ok = ok and torch.equal(parallel.gather_slabs(full2[:, b:e].contiguous(), 75), full2)
    with open(os.path.join(out_dir, f"ok8_{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_world_8_tile_shard_and_slab_gather_with_the_real_cost_vector(tmp_path):
    """VERDICT r4 #4a: world sizes > 4 were covered by a hand-run dry run only."""
    costs = parallel.tile_costs_host((160, 384, 384), (32, 128, 128), (8, 16, 16), (8, 16, 16))
    assert len(costs) == 160 and abs(sum(costs) / 1e12 - 68.86) < 0.01            # the frame-aware TFLOP per volume of DESIGN section 3
    assert min(costs) < 0.65 * max(costs) and costs[0] == min(costs)
    mp.spawn(_worker8, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    assert all((tmp_path / f"ok8_{r}").read_text() == "1" for r in range(8))


def _worker_queue(rank, world, port, n, out_dir):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q = parallel.VolumeQueue(n)
    got = []
    for i in q:
        got.append(i)
        time.sleep(0.02 * (1 + 3 * (rank == 0)))                 # rank 0 is four times slower: it must end up with fewer volumes
    assert q.claim() is None
    q2 = parallel.VolumeQueue(5)                                  # a second queue gets its own key on every rank
    got2 = list(q2)
    assert q.claimed == got and q2.claimed == got2                 # what this rank took is recorded (a driver can re-issue lost volumes)
    # a queue that the ranks construct with different sizes (e.g. in rank-conditional code) is refused instead of silently skipping volumes
    dist.barrier()
    try:
        parallel.VolumeQueue(7 if rank == 0 else 8, name="mismatched")
        ok = rank == 0 or None                                    # the rank that creates the key cannot know; every other rank must raise
    except RuntimeError:
        ok = True
    dist.barrier()
    created_first = int(dist.distributed_c10d._get_default_store().get("mismatched_n"))
    assert ok is True or created_first == (7 if rank == 0 else 8), "a size mismatch went unnoticed"
    with open(os.path.join(out_dir, f"q_{rank}"), "w") as f:
        f.write(",".join(map(str, got)) + ";" + ",".join(map(str, got2)))
    dist.barrier()
    dist.destroy_process_group()


def test_volume_queue_hands_out_every_volume_once_and_balances(tmp_path):
    world, n = 3, 40
    mp.spawn(_worker_queue, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    firsts, seconds = [], []
    for r in range(world):
        a, b = (tmp_path / f"q_{r}").read_text().split(";")
        firsts.append([int(v) for v in a.split(",") if v])
        seconds.append([int(v) for v in b.split(",") if v])
    assert sorted(sum(firsts, [])) == list(range(n)) and sorted(sum(seconds, [])) == list(range(5))
    assert len(firsts[0]) < min(len(firsts[1]), len(firsts[2]))           # the slow rank claimed fewer
    q = parallel.VolumeQueue(3)                                           # no process group: a local counter
    assert list(q) == [0, 1, 2] and q.claim() is None


# ---- one fp16x3 calibration per cohort: every rank leaves with rank 0's exponents ----------------------------

class _FakeEngine:
    """The members of UNetEngine that parallel.sync_calibration touches."""
    weights_sha256 = "abc"

    def __init__(self):
        self.exps, self.cal, self.calls = [0] * 18, False, 0
        self.refused, self.no_census, self.effective_precision = False, False, "fp16x3"

    def act_exponents(self):
        return list(self.exps), self.cal

    def set_act_exponents(self, e):
        self.exps, self.cal = [int(v) for v in e], True

    def calibration_status(self):
        return "refused_f32" if self.refused else "no_census" if self.no_census else "calibrated" if self.cal else "uncalibrated"

    def refuse_fp16(self, reason=""):
        self.refused, self.effective_precision = True, "f32"

    def mark_no_census(self):
        self.no_census = True


def _cal_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _FakeEngine()

    def calibrate():                        # what each rank WOULD arrive at on its own first volume: different per rank
        eng.calls += 1
        eng.set_act_exponents([rank + 3] * 17 + [0])

    got = parallel.sync_calibration(eng, calibrate)
    # a second cohort on an engine that is calibrated already (e.g. from the checkpoint's sidecar): nobody calibrates again
    got2 = parallel.sync_calibration(eng, calibrate)
    np.save(os.path.join(out_dir, f"cal_{rank}.npy"), np.asarray([got, got2, [eng.calls] * 18]))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_calibration_gives_every_rank_rank0s_exponents(tmp_path):
    world = 3
    mp.spawn(_cal_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"cal_{r}.npy") for r in range(world)]
    for r in range(world):
        assert res[r][0].tolist() == [3] * 17 + [0] and res[r][1].tolist() == [3] * 17 + [0]
        assert res[r][2][0] == (1 if r == 0 else 0)           # only rank 0 ran a calibration, once
    # single process, no group: calibrates when needed, not when the engine already holds exponents
    eng = _FakeEngine()
    assert parallel.sync_calibration(eng, lambda: eng.set_act_exponents([5] * 17 + [0])) == [5] * 17 + [0]
    assert parallel.sync_calibration(eng, lambda: eng.set_act_exponents([9] * 18)) == [5] * 17 + [0]


def _cal_status_worker(rank, world, port, out_dir, scenario):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _FakeEngine()

    def calibrate():
        eng.calls += 1
        if scenario == "refused":
            eng.refuse_fp16("did not settle")           # what UNetEngine.calibrate does when the exponents do not settle
        elif scenario == "no_census":
            eng.mark_no_census()
        elif scenario == "error":
            raise ValueError("volume 0 is unreadable")
        elif scenario == "silent":
            pass                                          # a calibrate_fn that returns without a verdict

    err = ""
    try:
        parallel.sync_calibration(eng, calibrate)
        if scenario == "no_census":
            parallel.sync_calibration(eng, calibrate)     # a second cohort: nobody calibrates again (one source of "calibrated?")
    except (RuntimeError, ValueError) as exc:
        err = type(exc).__name__ + ": " + str(exc)
    with open(os.path.join(out_dir, f"st_{rank}"), "w") as f:
        f.write(f"{eng.calibration_status()}|{eng.effective_precision}|{eng.calls}|{err}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["refused", "no_census", "error", "silent"])
def test_sync_calibration_propagates_rank0s_outcome(tmp_path, scenario):
    """a refused calibration on rank 0 must put EVERY rank on f32 (not rank 0 on f32 and the others on fp16x3 with
    exponents of their own first volume); a network without a census is not re-calibrated by every cohort; a calibrate_fn that raises on
    rank 0 -- or returns without a verdict -- publishes an error instead of leaving the other ranks in store.wait until the timeout."""
    world = 3
    mp.spawn(_cal_status_worker, args=(world, _free_port(), str(tmp_path), scenario), nprocs=world, join=True)
    res = [(tmp_path / f"st_{r}").read_text().split("|") for r in range(world)]
    for r, (status, prec, calls, err) in enumerate(res):
        assert int(calls) == (1 if r == 0 else 0)
        if scenario == "refused":
            assert status == "refused_f32" and prec == "f32" and err == ""
        elif scenario == "no_census":
            assert status == "no_census" and prec == "fp16x3" and err == ""
        else:
            assert err and ("rank 0" in err or r == 0), err
            assert ("unreadable" in err) if scenario == "error" else ("uncalibrated" in err)


# ---- a dropped calibration FILE: the ranks of a volume-parallel cohort agree on ONE recalibration -----------------

def test_calibration_board_one_rank_recalibrates_and_every_rank_mirrors_it():
    """Three "ranks" (threads over one HashStore -- the board only touches the store) whose engines run a file calibration.  Rank 1 hits the
    drop first: it claims the epoch, calibrates on ITS volume and publishes; rank 2 drops later, finds the epoch claimed and mirrors instead
    of calibrating; rank 0 never drops and picks the publication up at its next poll.  Everybody ends on rank 1's exponents, exactly one
    calibration ran; a board without a store (single process) just calibrates."""
    import threading
    store = dist.HashStore()
    engines = [_FakeEngine() for _ in range(3)]
    for e in engines:
        e.set_act_exponents([1] * 17 + [0])                     # "from the sidecar"
    boards = [parallel.CalibrationBoard(store, name="t") for _ in range(3)]
    assert not boards[0].poll(engines[0])                       # nothing published yet: non-blocking, nothing changes
    ran = []

    def cal(rank):
        def fn():
            ran.append(rank)
            engines[rank].set_act_exponents([rank + 4] * 17 + [0])
        return fn

    gate = threading.Event()
    out = {}

    def late_rank():                                            # rank 2 drops while rank 1 is still calibrating: it must WAIT, then mirror
        gate.wait()
        out[2] = boards[2].recalibrate(engines[2], cal(2))

    th = threading.Thread(target=late_rank)
    th.start()

    def slow_cal():
        gate.set()                                              # rank 2 arrives while the epoch is claimed but not published
        import time
        time.sleep(0.2)
        cal(1)()

    out[1] = boards[1].recalibrate(engines[1], slow_cal)
    th.join(timeout=30)
    assert not th.is_alive()
    assert out == {1: True, 2: False} and ran == [1]
    assert boards[0].poll(engines[0]) and not boards[0].poll(engines[0])
    for e in engines:
        assert e.act_exponents() == ([5] * 17 + [0], True)
    assert [b.epoch for b in boards] == [1, 1, 1]
    # a rank that drops after it has (unknowingly) been superseded mirrors at once
    solo = parallel.CalibrationBoard(None)
    e = _FakeEngine()
    assert solo.recalibrate(e, lambda: e.set_act_exponents([2] * 17 + [0])) and e.act_exponents()[0][0] == 2


def test_calibration_board_publishes_a_failure_instead_of_leaving_ranks_waiting():
    store = dist.HashStore()
    a, b = parallel.CalibrationBoard(store, name="f"), parallel.CalibrationBoard(store, name="f")
    ea, eb = _FakeEngine(), _FakeEngine()

    def boom():
        raise OSError("volume unreadable")

    with pytest.raises(OSError):
        a.recalibrate(ea, boom)
    with pytest.raises(RuntimeError, match="unreadable"):
        b.recalibrate(eb, lambda: eb.set_act_exponents([3] * 18))     # the epoch is claimed and published as an error: no hang, no own calibration
    assert eb.act_exponents()[1] is False
    c, ec = parallel.CalibrationBoard(store, name="g"), _FakeEngine()
    with pytest.raises(RuntimeError, match="uncalibrated"):
        c.recalibrate(ec, lambda: None)                                # returned without a verdict: published as an error too

