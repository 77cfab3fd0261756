"""GPU: the process-group path on real hardware -- one rank under torch.distributed.run with the "nccl" (RCCL) backend, the only
multi-process configuration a one-GPU box allows.  The CPU/gloo tests (tests/test_parallel_cpu.py) cover world sizes 2-4; this one
covers what they cannot: RCCL initialisation next to liboai_hip.so in the same process, every collective of parallel.py on DEVICE
tensors, and run_sharded == run under an initialised group."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_run_sharded_equals_run_under_an_rccl_process_group_of_one():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker_gpu.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)      # a CHILD process: this one keeps its GPU context
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["world"] == 1 and out["backend"] == "nccl"
    assert out["calibrated"] and len(out["exponents"]) == 18
    assert out["equal"], "run_sharded differs from run under the process group"
    assert out["flag"] == 0 and out["fc_sum"] > 0


def test_run_sharded_at_world_2_on_one_gpu_over_gloo():
    """the sharded path had never run between two processes on GPU memory (the box has one MI355X and RCCL does not put two
    ranks on one device).  Two ranks on device 0 with the gloo backend on DEVICE tensors: broadcast of the volume, cost-balanced tile ranges
    computed straight into the gather buffer's slots, the in-place ragged all_gather, ``oai_stitch_blocks_ranged`` over a two-range slot table,
    the MAX-reduced range state, the z-slab gather -- and every rank's ``run_sharded`` result equals its own ``run``."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", OAI_TEST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker_gpu.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["world"] == 2 and out["backend"] == "gloo"
    assert out["tile_ranges"][0][1] == out["tile_ranges"][1][0] and out["tile_ranges"][1][1] == out["n_tiles"] == 75
    assert out["equal"] and out["equal_on_every_rank"], "run_sharded differs from run at world 2"
    assert out["flag"] == 0 and out["fc_sum"] > 0


def test_process_cohort_at_world_2_on_one_gpu_over_gloo(tmp_path):
    """``dask_processing.process_cohort`` -- the driver that replaces the reference's Dask graph (dask_processing.py:46-189) -- between two real
    processes: one shared volume queue (every volume claimed exactly once), one calibration for the group through the store, a streaming
    CohortRunner per rank; and a volume's results do not depend on the rank that claimed it: each equals the same volume run alone on rank 0."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", OAI_TEST_DIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_cohort_worker_gpu.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["world"] == 2 and out["calibrated"] and out["same_calibration"]
    claimed = sorted(i for c in out["claimed"] for i in c)
    assert claimed == list(range(7)), out["claimed"]                       # every volume exactly once, whatever the split
    assert all(out["match_alone"].values()), out["match_alone"]

