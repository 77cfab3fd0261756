"""What do 8 processes' worth of CohortRunner host legs cost one host?  (VERDICT r5 #4b; reference loop: dask_processing.py:170-181, one worker per GPU)

An 8-GPU cohort at 7.6 volumes/s per GPU moves, per process and second, 7.6 x 94 MB pageable -> pinned (the upload worker's staging copy) and
7.6 x 566 MB pinned -> freshly allocated pageable memory the caller owns (the download worker's five side-by-side copies): 8 x 7.6 x 0.66 GB = 40 GB/s
of host memcpy plus the page faults of 34 GB/s of fresh memory, on ONE host.  This script starts N child processes that run exactly those legs (the
same staging pattern, thread pools and buffer counts as cohort.CohortRunner) around a STUB device leg (nothing crosses PCIe: the box has one GPU, the
question is the host), in two regimes:

    paced    every process is offered one volume per 1 / 7.6 s (what 8 GPUs would ask of the host): does each keep up?  (achieved volumes/s, worker busy share)
    flat out no pacing: the host's ceiling for these legs (aggregate GB/s), and how much of a copy's time is page faults (fresh vs pre-faulted destinations)

Usage: python scripts/host8.py [N=8] [volumes=24]     (prints one JSON line per regime and N; pinned memory needs a HIP context: each child creates one)
"""
import json
import multiprocessing as mp
import os
import sys
import time

SHAPES = [(160, 384, 384), (160, 384, 384), (3, 80, 192, 192), (160, 384, 384), (160, 384, 384)]       # fc, tc, phi, fc_atlas, tc_atlas (float32)
VOL = (160, 384, 384)
RATE = 7.6                                                                                            # volumes/s per GPU (BENCH_r05: 7.59)


def worker(rank, n_vol, paced, barrier, out_q, recycle=0):
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor
    torch.set_num_threads(1)
    torch.cuda.set_device(0)
    pin_in = [torch.empty(VOL, dtype=torch.float32).pin_memory() for _ in range(2)]
    pin_out = [[torch.empty(s, dtype=torch.float32).pin_memory() for s in SHAPES] for _ in range(3)]     # CohortRunner.N_OUT_SETS
    for s in pin_out:
        for t in s:
            t.fill_(1.0)
    src = [np.random.rand(*VOL).astype(np.float32) for _ in range(2)]                                    # the caller's pageable volumes
    up, clone = ThreadPoolExecutor(1), ThreadPoolExecutor(5)
    stats = {"stage_s": 0.0, "stage_b": 0, "clone_s": 0.0, "clone_b": 0, "clone_prefaulted_s": 0.0}

    def stage(i):
        t0 = time.perf_counter()
        pin_in[i & 1].copy_(torch.from_numpy(src[i & 1]))
        stats["stage_s"] += time.perf_counter() - t0
        stats["stage_b"] += pin_in[0].numel() * 4

    ring = [None] * recycle                                                                              # CohortRunner(result_pool=recycle): recycled, pre-faulted result sets

    def collect(i):
        pins = pin_out[i % 3]
        t0 = time.perf_counter()
        if recycle:
            if ring[i % recycle] is None:
                ring[i % recycle] = [torch.empty(p.shape, dtype=p.dtype) for p in pins]
            outs = list(clone.map(lambda pd: pd[1].copy_(pd[0]), zip(pins, ring[i % recycle])))
        else:
            outs = list(clone.map(lambda p: torch.empty(p.shape, dtype=p.dtype).copy_(p), pins))       # fresh pages: faulted inside the copy
        stats["clone_s"] += time.perf_counter() - t0
        stats["clone_b"] += sum(p.numel() * 4 for p in pins)
        return outs

    from collections import deque
    down = ThreadPoolExecutor(1)
    keep, pending = None, deque()
    barrier.wait()
    t_start = time.perf_counter()
    for i in range(n_vol):
        if paced:
            due = t_start + i / RATE
            now = time.perf_counter()
            if now < due:
                time.sleep(due - now)
        f_up = up.submit(stage, i)                        # upload worker: pageable -> pinned (the H2D behind it is the stub)
        pending.append(down.submit(collect, i))           # download worker: pinned set i % 3 -> fresh caller-owned tensors (the D2H in front of it is the stub)
        while len(pending) > 2:                           # CohortRunner.LAG: results are handed out two volumes behind
            keep = pending.popleft().result()
        f_up.result()
    while pending:
        keep = pending.popleft().result()
    dt = time.perf_counter() - t_start
    # page-fault share: the same copy into destinations that are already faulted in
    t0 = time.perf_counter()
    for _ in range(3):
        list(clone.map(lambda pd: pd[1].copy_(pd[0]), zip(pin_out[0], keep)))
    stats["clone_prefaulted_s"] = (time.perf_counter() - t0) / 3
    out_q.put({"rank": rank, "volumes_per_s": n_vol / dt, "seconds": dt,
               "stage_GBps": stats["stage_b"] / stats["stage_s"] / 1e9, "clone_GBps": stats["clone_b"] / stats["clone_s"] / 1e9,
               "clone_ms_per_volume": 1e3 * stats["clone_s"] / n_vol, "clone_prefaulted_ms_per_volume": 1e3 * stats["clone_prefaulted_s"],
               "stage_ms_per_volume": 1e3 * stats["stage_s"] / n_vol,
               "download_worker_busy": stats["clone_s"] / dt, "upload_worker_busy": stats["stage_s"] / dt})


def run(n_proc, n_vol, paced, recycle=0):
    ctx = mp.get_context("spawn")
    barrier, q = ctx.Barrier(n_proc), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, n_vol, paced, barrier, q, recycle)) for r in range(n_proc)]
    for p in ps:
        p.start()
    res = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join()
    import math
    bytes_out = sum(4 * math.prod(s) for s in SHAPES)
    bytes_in = 4 * VOL[0] * VOL[1] * VOL[2]
    agg = sum(r["volumes_per_s"] for r in res)
    return {"processes": n_proc, "regime": ("paced at 7.6 volumes/s per process" if paced else "flat out") + (f", results in {recycle} recycled sets" if recycle else ""), "volumes_per_process": n_vol,
            "aggregate_volumes_per_s": agg, "aggregate_host_memcpy_GBps": agg * (bytes_in + bytes_out) / 1e9,
            "per_process_volumes_per_s": {"min": min(r["volumes_per_s"] for r in res), "max": max(r["volumes_per_s"] for r in res)},
            "clone_GBps_per_process": {"min": min(r["clone_GBps"] for r in res), "max": max(r["clone_GBps"] for r in res)},
            "stage_GBps_per_process": {"min": min(r["stage_GBps"] for r in res), "max": max(r["stage_GBps"] for r in res)},
            "clone_ms_per_volume": max(r["clone_ms_per_volume"] for r in res), "clone_prefaulted_ms_per_volume": max(r["clone_prefaulted_ms_per_volume"] for r in res),
            "page_fault_share_of_clone": 1.0 - min(r["clone_prefaulted_ms_per_volume"] / r["clone_ms_per_volume"] for r in res),
            "stage_ms_per_volume": max(r["stage_ms_per_volume"] for r in res),
            "download_worker_busy_max": max(r["download_worker_busy"] for r in res), "upload_worker_busy_max": max(r["upload_worker_busy"] for r in res),
            "host_cores": os.cpu_count()}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    vols = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    for n_proc in sorted({1, n}):
        for paced in (True, False):
            print(json.dumps(run(n_proc, vols, paced)), flush=True)
    for paced in (True, False):                       # CohortRunner(result_pool=4): the download leg without its page faults
        print(json.dumps(run(n, vols, paced, recycle=4)), flush=True)
