#!/usr/bin/env python
"""bench.py -- knee MRI volumes/s (segment + register + FC/TC resample), 384x384x160 fp32.

One "step" = one synthetic DESS volume through the whole per-volume hot path on one GPU:
overlap-tiled 3D U-Net segmentation (160 tiles of 128x128x32, reference tiling) -> stitch -> ICON
registration to the atlas (one direction, what ICON_Registration.register returns) -> both probability
maps pulled onto the atlas grid through phi.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W          (N > 1: spawns its N ranks itself, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU over RCCL.  --mode replicas (default): volumes are independent units (the reference's own
Dask model), every rank processes its own volume per step, no data-path collective ("scaling": "weak").
--mode tileshard: every step is ONE volume: broadcast from rank 0, its 160 tiles split over the ranks, one all_gather,
the phi-resample sharded by atlas z-slab ("scaling": "strong").  Rank 0 prints ONE JSON line.

--dry-run: no GPU, no kernels: the launcher, the rendezvous and the collectives of oai_analysis_2_amd.parallel run on CPU
tensors over gloo with a stand-in per-rank compute (CI check of the multi-process plumbing; "value" is null).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VOL_SHAPE = (160, 384, 384)
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32
MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense 16-bit MFMA
WINO_F32_EXECUTED = 2.0 / 3.0     # conv3_wino_f32: MFMAs executed per algorithmic MFMA (every k3 layer of the exact-fp32 path; ec0 has its own kernel)
PASSES = {"f32": 1, "bf16x6": 6, "bf16x3": 3, "fp16x3": 3}
SUSTAINED_16BIT_MFMA_TFLOPS = 1857.0   # scripts/micro/mfma_peak.hip on this chip: operands in registers, every CU (profiles/r01_ablation.md)
KERNEL_OF = {"f32": "conv3_wino_f32 (exact fp32 products, x axis in Winograd F(2,3) form: 2/3 of the direct form's MFMAs; conv3_igemm_f32 with option winograd_f32=0)", "fp16x3": "conv3_wino_sres / conv3_igemm_sres (split-resident fp16x3; x axis of the plain layers in Winograd F(2,3) form)",
             "bf16x3": "conv3_igemm_bf16s (split 16-bit)", "bf16x6": "conv3_igemm_bf16s (split 16-bit)"}
DTYPE_OF = {"f32": "f32", "bf16x6": "bf16x6 (fp32 operands split into 3 bf16 terms, 6 MFMA passes, fp32 accumulate)",
            "bf16x3": "bf16x3 (2 bf16 terms, 3 MFMA passes, fp32 accumulate)",
            "fp16x3": "fp16x3 (every fp32 value held as 2 fp16 terms = 22 mantissa bits, 3 MFMA passes per product, fp32 accumulate; "
                      "fp32 in and out of every entry point; full-size parity vs the reference in `parity`)"}
TRAFFIC_FILE = {"f32": "profiles/r06_pmc_traffic_f32.json", "fp16x3": "profiles/r06_pmc_traffic_sres.json"}


def physical_cores() -> int:
    """Physical cores of this host (unique (package, core) pairs of /proc/cpuinfo); BASELINE.md 3 asks for this, not SMT threads."""
    try:
        pairs, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of torch.distributed.run and return
    their exit code.  Nothing here touches the GPU (a process that has initialised HIP must not exec or fork GPU work)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(vol_np, meta_A, atlas_img, unet_sd, icon_sd, n_tiles_sample=8, reps=3):
    """The oracle (CPU port of the reference algorithm) timed on this host's physical cores, on a bounded sample, with BASELINE.md 3's
    protocol: 8 tiles (x20), fed 4 at a time like the reference's loop (analysis_object.py:22, segmenter.py:109-119), 1 warm-up +
    3 timed repetitions; the registration and a tenth of one resample once each."""
    import numpy as np
    import torch
    from oai_analysis_2_amd.image import Image
    from oracle import icon as oicon, resample as oresample, seg as oseg
    cores = physical_cores()
    torch.set_num_threads(cores)
    tiles, g = oseg.partition(vol_np, (128, 128, 32), (16, 16, 8))
    x = torch.from_numpy(np.ascontiguousarray(tiles[:n_tiles_sample]))
    oseg.unet_forward(x[:4], unet_sd)                                    # warm-up
    rep_s = []
    for _ in range(reps):
        t0 = time.time()
        for i in range(0, n_tiles_sample, 4):
            oseg.unet_forward(x[i:i + 4], unet_sd)
        rep_s.append(time.time() - t0)
    t_tile = min(rep_s) / n_tiles_sample
    t0 = time.time()
    phi, _ = oicon.register_pair_arrays(vol_np, atlas_img.array, icon_sd, both=False)
    t_reg = time.time() - t0
    disp = oicon.displacement_itk(phi)
    t0 = time.time()
    zs = 16                                                               # 16 of 160 atlas slices, scaled x10
    sub = Image(atlas_img.array[:zs], atlas_img.spacing, atlas_img.origin, atlas_img.direction)
    oresample.resample_through_phi(vol_np.astype(np.float64), disp, meta_A, sub)
    t_res = (time.time() - t0) * (atlas_img.array.shape[0] / zs) * 2      # FC and TC
    t_vol = g["n_tiles"] * t_tile + t_reg + t_res
    return {"value": 1.0 / t_vol, "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": f"{n_tiles_sample} of {g['n_tiles']} U-Net tiles in batches of 4 (x{g['n_tiles'] / n_tiles_sample:.0f}), 1 warm-up + {reps} repetitions "
                      f"(best; all: {', '.join('%.1f' % r for r in rep_s)} s), 1 full ICON direction, {zs}/{atlas_img.array.shape[0]} slices of one resample "
                      f"(x{2 * atlas_img.array.shape[0] // zs}); s/tile={t_tile:.2f} s_register={t_reg:.1f} s_resample={t_res:.1f}; torch threads = physical cores for the "
                      f"U-Net and ICON legs; the resample leg is oracle/resample.py, SINGLE-THREADED numpy float64, where the reference runs ITK's multithreaded "
                      f"ResampleImageFilter: that leg ({100 * t_res / t_vol:.0f} % of the per-volume time) is over-stated, i.e. this baseline is UNDER-stated by at most that share"}


def fullsize_parity(unet, precision, case="base"):
    """The segmentation of a golden volume in `precision` against tests/golden/segment_fullsize[_<case>].npz -- the REFERENCE's own
    segment() run on CPU at 384x384x160 (tests/golden/make_golden_fullsize.py --case ...; `unet` must hold that case's network).
    Data fixture, not the oracle; outside the timed region.  The same comparison is asserted by tests/test_fullsize_gpu.py."""
    import hashlib
    import numpy as np
    import torch
    from oai_analysis_2_amd.pipeline import CROP_ZYX, OVERLAP_ZYX, TILE_ZYX
    from oai_analysis_2_amd.synth import FULLSIZE_CASES, make_volume, make_volume_windowed
    c = FULLSIZE_CASES[case]
    path = os.path.join(ROOT, "tests", "golden", c["file"])
    if not os.path.exists(path):
        return None
    z = np.load(path)
    vol = (make_volume_windowed if c["windowed"] else make_volume)(int(z["volume_seed"]), VOL_SHAPE)
    same_input = hashlib.sha256(vol.tobytes()).digest() == bytes(z["volume_sha256"])
    prev = unet.precision
    unet.set_precision(precision)
    v = torch.from_numpy(vol).cuda()
    prob = unet.stitch(unet.segment_tiles(v, TILE_ZYX, OVERLAP_ZYX, out_mode=0, crop_zyx=CROP_ZYX), VOL_SHAPE, TILE_ZYX, OVERLAP_ZYX, CROP_ZYX)
    mask = unet.stitch(unet.segment_tiles(v, TILE_ZYX, OVERLAP_ZYX, out_mode=1, crop_zyx=CROP_ZYX), VOL_SHAPE, TILE_ZYX, OVERLAP_ZYX, CROP_ZYX)
    unet.set_precision(prev)
    prob, mask = prob.cpu().numpy(), mask.cpu().numpy() > 0.5
    ref_mask = np.stack([np.unpackbits(z["fc_mask_bits"])[:vol.size], np.unpackbits(z["tc_mask_bits"])[:vol.size]]).astype(bool).reshape(2, *VOL_SHAPE)
    flips = np.flatnonzero((mask != ref_mask).ravel())
    near = dict(zip(z["near_idx"].tolist(), z["near_prob"].tolist()))
    worst = max((abs(near.get(int(i), 0.0) - 0.5) for i in flips), default=0.0)
    sl = tuple(slice(int(a), None, int(s)) for a, s in zip(z["start"], z["stride"]))
    ref_s = np.stack([z["fc_prob_s"], z["tc_prob_s"]]).astype(np.float64)
    got_s = np.stack([prob[0][sl], prob[1][sl]]).astype(np.float64)
    dsum = np.abs(got_s - ref_s).sum(axis=(1, 2, 3))
    scale = vol.size / ref_s[0].size
    return {"against": f"tests/golden/{c['file']} = the reference's Segmenter3DInPatchClassWise.segment on CPU, 384x384x160, seeded weights",
            "precision": precision, "input_bit_identical": bool(same_input), "mask_voxels": int(2 * vol.size),
            "mask_flips": int(len(flips)), "max_abs_pref_minus_half_at_flips": float(worst),
            "flips_with_pref_farther_than_1e-5_from_half": int(sum(1 for i in flips if abs(near.get(int(i), 0.0) - 0.5) >= 1e-5)),
            "sum_abs_dp_per_23.6M_voxels": [float(d * scale) for d in dsum], "reference_budget_sum_abs_dp": 12.0,
            "max_abs_dp_sample": float(np.abs(got_s - ref_s).max()), "sample_voxels_per_map": int(ref_s[0].size)}


def other_cases_parity(precision):
    """the same comparison on three more reference runs (other weight seeds, BN=True, DC-heavy activations, an
    intensity-windowed input; oai_analysis_2_amd.synth.FULLSIZE_CASES), each with its own engine and calibration; compact."""
    import torch
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    from oai_analysis_2_amd.synth import make_fullsize_case
    out = {}
    try:                                                           # the reference's OWN fp32-vs-fp64 distance per network (tests/golden/make_golden_truth.py)
        import numpy as np
        truth = np.load(os.path.join(ROOT, "tests", "golden", "segment_fullsize_truth.npz"))
    except OSError:
        truth = None
    for case in ("bn", "dc", "win"):
        sd, _, c = make_fullsize_case(case, (8, 8, 8))             # (the volume is regenerated at full size inside fullsize_parity)
        eng = UNetEngine(sd, precision=precision)
        r = fullsize_parity(eng, precision, case)
        if r is not None:
            out[case] = {"network": f"weight seed {c['weight_seed']}, BN={c['bn']}, bias shift {c['bias_shift']}",
                         "volume": f"seed {c['volume_seed']}" + (", 5 % of the voxels at exactly 0 and at exactly 1" if c["windowed"] else ""),
                         "input_bit_identical": r["input_bit_identical"], "mask_flips": r["mask_flips"],
                         "flips_with_pref_farther_than_1e-5_from_half": r["flips_with_pref_farther_than_1e-5_from_half"],
                         "sum_abs_dp_per_23.6M_voxels": r["sum_abs_dp_per_23.6M_voxels"], "max_abs_dp_sample": r["max_abs_dp_sample"],
                         "reference_own_fp32_noise_sum_abs_dp": truth[f"{case}_ref_err"].tolist() if truth is not None else None,
                         "range_flag": eng.range_flag() if precision == "fp16x3" else 0}
        eng._ws = None
        del eng
        torch.cuda.empty_cache()
    return out


def icon_step_tree_times(A, B):
    """Device time of one registration direction at 80x192x192 for the step trees a checkpoint's keys may spell:
    today's three steps and the four-step form with a second full-resolution U-Net.  Outside the timed region."""
    import torch
    from oai_analysis_2_amd.registration import IconEngine
    from oai_analysis_2_amd.synth import make_icon_state_dict
    out = {}
    for tree in ("3step", "4step"):
        eng = IconEngine(make_icon_state_dict(0, 0.1, tree))
        for _ in range(2):
            eng.phi(A, B)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.phi(A, B)
        e1.record()
        torch.cuda.synchronize()
        out[tree] = {"tree": eng.tree.describe(), "ms_per_direction": e0.elapsed_time(e1) / 10}
    return out


def rescaled_network_parity(unet_sd):
    """the golden network with ec0.{weight,bias} * 2^-10, ec1.weight * 2^10, dc5.{weight,bias} * 2^-14, dc4.weight * 2^14 is
    the SAME fp32 function (powers of two commute with ReLU and every rounding), so the reference golden applies unchanged; its
    activations at ec0 / dc5 sit 2^10 / 2^14 below the original's.  fp16x3 with calibrated activation exponents must give the same
    parity numbers as on the original network (tests/test_fp16_range_gpu.py asserts it; uncalibrated: 4e-4 relative logit error)."""
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    sd = {k: v.clone() for k, v in unet_sd.items()}
    for prod, k, cons in (("ec0", 10, "ec1"), ("dc5", 14, "dc4")):
        sd[f"{prod}.0.weight"] *= 2.0 ** -k
        sd[f"{prod}.0.bias"] *= 2.0 ** -k
        sd[f"{cons}.0.weight"] *= 2.0 ** k
    eng = UNetEngine(sd, precision="fp16x3")
    out = fullsize_parity(eng, "fp16x3")
    if out is not None:
        out["network"] = "golden weights with ec0 * 2^-10 -> ec1 * 2^10 and dc5 * 2^-14 -> dc4 * 2^14 (bit-identical fp32 function)"
        out["activation_exponents"] = eng.act_exponents()[0]
        out["range_flag"] = eng.range_flag()
    import torch
    eng._ws = None                                                        # hand the second engine's activation workspace back before the next leg
    del eng
    torch.cuda.empty_cache()
    return out


def dry_run(args, world, rank):
    """CPU/gloo check of the multi-process plumbing: broadcast, cost-balanced tile shard + all_gather, flag agreement,
    z-slab gather, max-over-ranks timing -- the product's own parallel.py on CPU tensors, with a stand-in compute."""
    import torch
    import torch.distributed as dist
    from oai_analysis_2_amd import parallel
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    shape = (8, 12, 12)
    # the BASELINE geometry's own split: 160 tiles with the trimmed per-tile costs (border tiles up to 40 % cheaper: 23 / 19 x 6 / 23 tiles at
    # 8 ranks -- ragged ranges through the in-place gather + the stitch's slot table) and the 160 atlas slices of the slab-sharded resample
    n_tiles, nz = 160, 160
    costs = parallel.tile_costs_host(VOL_SHAPE, (32, 128, 128), (8, 16, 16), (8, 16, 16))
    vol = torch.arange(8 * 12 * 12, dtype=torch.float32).reshape(shape) if rank == 0 else None
    t0 = time.perf_counter()
    for _ in range(args.warmup + args.steps):
        if args.mode == "tileshard":
            v = parallel.broadcast_volume(vol, shape, "cpu", 0) if world > 1 else vol
            def compute(rg, out):                                      # this rank's blocks, written straight into its slot of the gather buffer
                for j, t in enumerate(range(*rg)):
                    out[j] = float(v.sum()) + t
            g = parallel.segment_tile_sharded(compute, n_tiles, None, costs, block_shape=(2, 2, 3, 3), dtype=torch.float32, device="cpu")
            assert g.n_tiles == n_tiles and len(g.bounds) == world + 1
            for t in range(n_tiles):                                   # the slot table oai_stitch_blocks_ranged reads through
                r = max(i for i in range(world) if g.bounds[i] <= t)
                assert float(g.buffer[r * g.stride + t - g.bounds[r], 0, 0, 0, 0]) == float(v.sum()) + t
            blocks = g.compact()
            assert blocks.shape[0] == n_tiles and all(float(blocks[t, 1, 1, 2, 2]) == float(v.sum()) + t for t in range(0, n_tiles, 7))
            flag = parallel.any_rank(torch.tensor([1 if rank == world - 1 else 0], dtype=torch.int32))
            assert int(flag) == 1
            b, e = parallel.slab_range_for_rank(nz, rank, world)
            slabs = parallel.gather_slabs((torch.arange(b, e, dtype=torch.float32)[None, :, None, None] + 1000.0 * torch.arange(2)[:, None, None, None]).expand(2, e - b, 3, 3).contiguous(), nz)
            assert slabs.shape == (2, nz, 3, 3) and torch.equal(slabs[0, :, 0, 0], torch.arange(nz, dtype=torch.float32)) and torch.equal(slabs[1, :, 2, 1], 1000.0 + torch.arange(nz, dtype=torch.float32))
        elif args.mode == "cohort":
            q = parallel.VolumeQueue(4 * world)
            mine = list(q)
            total = torch.tensor([len(mine)], dtype=torch.int64)
            if world > 1:
                dist.all_reduce(total)
            assert int(total) == 4 * world and len(set(mine)) == len(mine)
            # the per-rank streamed leg of the GPU run: each rank's [steady ms, volumes/s, GB/s out of pinned, GB/s into pinned] gathered to every rank
            row = torch.tensor([130.0 + rank, 7.0, 40.0 + rank, 60.0], dtype=torch.float64)
            rows = [torch.zeros(4, dtype=torch.float64) for _ in range(world)]
            if world > 1:
                dist.all_gather(rows, row)
            else:
                rows = [row]
            streamed = aggregate_streamed([r.tolist() for r in rows], resident_ms=129.0)
            assert streamed["ranks"] == world and abs(streamed["steady_ms_per_volume"]["max"] - (130.0 + world - 1)) < 1e-9
            assert abs(streamed["aggregate_steady_volumes_per_s"] - sum(1e3 / (130.0 + r) for r in range(world))) < 1e-9
        else:
            mine = parallel.volumes_for_rank(world * 2, rank, world)
            assert mine == [rank, rank + world]
    if world > 1:
        dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "knee MRI volumes/sec (segment+register), 384x384x160 fp32", "value": None, "unit": "volumes/s",
                          "dry_run": True, "backend": "gloo" if world > 1 else None, "world_size": world, "n_gpus": 0,
                          "tile_ranges": [list(parallel.tile_range_for_rank(n_tiles, r, world, costs)) for r in range(world)] if args.mode == "tileshard" else None,
                          "streamed_from_host": streamed if args.mode == "cohort" else None,
                          "mode": args.mode, "steps": args.steps, "warmup": args.warmup, "seconds": float(dt)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def aggregate_streamed(per_rank, resident_ms=None):
    """The N > 1 form of `streamed_from_host` (--mode cohort): every rank streams its own volumes from ITS host memory through ITS GPU at the same
    time -- N upload workers, N download workers and N x 5 clone threads on one host, N GPUs' PCIe links -- and reports [steady-state ms per
    volume, volumes/s incl. fill and drain, out-of-pinned GB/s, into-pinned GB/s]; this is what rank 0 prints from the gathered rows.  The
    aggregate is the SUM of the ranks' steady-state rates (independent units, no collective: SURVEY 8e), never `value`."""
    steady = [r[0] for r in per_rank]
    agg = sum(1e3 / ms for ms in steady if ms > 0)
    return {"ranks": len(per_rank), "aggregate_steady_volumes_per_s": agg, "aggregate_volumes_per_s_incl_fill_drain": sum(r[1] for r in per_rank),
            "steady_ms_per_volume": {"min": min(steady), "max": max(steady)},
            "vs_resident": (resident_ms / max(steady)) if resident_ms else None,
            "host_memcpy_GBps_per_rank": {"out_of_pinned_min": min(r[2] for r in per_rank), "into_pinned_min": min(r[3] for r in per_rank)},
            "what": "every rank: volumes in pageable host memory -> pinned staging -> H2D -> segment + register + resample -> D2H -> caller-owned host tensors "
                    "(cohort.CohortRunner), all ranks at once; resident = this run's ms_per_step"}


def streamed_from_host(pipe, n_volumes=24, resident_ms=None):
    """BASELINE config 4, PCIe inclusive: `n_volumes` synthetic volumes in HOST memory streamed through one GPU by CohortRunner (host staging
    + H2D of i+1, D2H of i-1's five result tensors and their copy into caller-owned memory overlap the compute of i).  Never `value`.
    Reported separately: FILL (start -> first result: one upload + one volume + its download, nothing overlapped),
    STEADY STATE (results 3 .. n-3: the inter-result interval, what a long cohort sees) and DRAIN; and the host memcpy rates of the two
    worker threads (566 MB per volume out of pinned memory, 94 MB into it)."""
    import torch
    from oai_analysis_2_amd.cohort import CohortRunner
    from oai_analysis_2_amd.image import Image
    from oai_analysis_2_amd.synth import make_volume
    base = [make_volume(i, VOL_SHAPE) for i in range(8)]                    # seeds 0..7, cycled: 24 volumes of host memory would be 2.3 GB of generation time
    imgs = [Image(base[i % 8], [0.36, 0.36, 0.7], [2.0, -3.0, 1.0]) for i in range(n_volumes)]
    runner = CohortRunner(pipe)
    for _ in runner.run(imgs[:3]):                                        # warm-up: pinned buffers, allocator, worker threads
        pass
    torch.cuda.synchronize()
    for k in runner.stats:
        runner.stats[k] = 0 if isinstance(runner.stats[k], int) else 0.0
    t0 = time.perf_counter()
    stamps, repeated = [], 0
    for _, r in runner.run(imgs):
        stamps.append(time.perf_counter() - t0)
        repeated += int(r.repeated_f32)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = len(stamps)
    lo, hi = 3, n - 3                                                     # steady state: away from the fill and the drain
    steady = (stamps[hi] - stamps[lo]) / (hi - lo) if hi > lo else None
    st = runner.stats
    runner.close()
    return {"value": n / dt, "unit": "volumes/s", "volumes": n, "seconds": dt, "repeated_in_f32": repeated,
            "fill_s": stamps[0], "drain_s": dt - stamps[hi] if hi < n else None,
            "steady_state": {"ms_per_volume": 1e3 * steady, "volumes_per_s": 1.0 / steady, "results": [lo, hi],
                             "vs_resident": (resident_ms / (1e3 * steady)) if (resident_ms and steady) else None} if steady else None,
            "host_memcpy": {"out_of_pinned_GBps": st["clone_bytes"] / st["clone_s"] / 1e9 if st["clone_s"] else None,
                            "into_pinned_GBps": st["stage_bytes"] / st["stage_s"] / 1e9 if st["stage_s"] else None,
                            "bytes_per_volume": (st["clone_bytes"] + st["stage_bytes"]) / max(n, 1),
                            "launch_thread_waited_s": st["launch_wait_s"],
                            "launch_thread_s": {k: round(st[k], 4) for k in ("t_upload_wait", "t_queue_compute", "t_issue_d2h", "t_result_wait") if k in st},
                            "note": "both copies run on worker threads (cohort.py); the launch thread only queues work"},
            "what": f"{n} volumes from pageable host arrays -> pinned staging -> H2D -> segment + register + resample -> D2H of fc, tc, phi, "
                    "fc_atlas, tc_atlas (566 MB per volume) -> caller-owned host tensors; upload / compute / download / host copy overlapped (cohort.CohortRunner)"}


def measure(step, unet, steps, warmup, use_dist, dist):
    """W untimed + K timed steps, barrier + synchronize on both sides; returns (seconds, conv ms, conv launches)."""
    import torch
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    unet.profile_read()
    unet.profile(True)                                                    # HIP events around the dominant kernel, on its stream
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flags = []
    for i in range(steps):
        flags.append(step(i).overflow)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    conv_ms, conv_launches = unet.profile_read()
    unet.profile(False)
    overflow = any(int(f.item()) for f in flags if f is not None)      # the per-volume fp16 range flags, read after the timed region
    return dt, conv_ms, conv_launches, overflow


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="tiles per U-Net pass (sizes the activation workspace); 0 = all that fit in 60%% of free HBM")
    ap.add_argument("--precision", default="fp16x3", choices=["fp16x3", "f32", "bf16x6", "bf16x3"],
                    help="arithmetic of the 3x3x3 conv layers.  fp16x3 (default): every fp32 operand split into two fp16 terms, "
                         "3 MFMA passes, fp32 accumulate -- fp32-grade results (same parity margins as f32 in tests/, incl. the full-size "
                         "reference golden); f32: exact fp32 MFMA; bf16x6 / bf16x3: split-bf16 with 6 / 3 passes")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "tileshard", "cohort"],
                    help="N>1: replicas = one volume per rank per step (weak scaling, no collective); tileshard = every step "
                         "is ONE volume whose 160 tiles are split over the ranks + one RCCL all_gather (strong scaling); cohort = "
                         "BASELINE config 5's form: --steps x N volumes (64 at --steps 8 --gpus 8) in ONE shared queue, every rank claims "
                         "the next volume when it is free (parallel.VolumeQueue) -- same work as replicas, dynamically assigned")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the second measurement in exact fp32 MFMA (full --steps) reported beside the primary")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size parity block (reference golden fixture)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the PCIe-inclusive leg (volumes streamed from host memory, BASELINE config 4; --mode cohort: per rank, all ranks at once)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="tuning option of the fp16x3 path (oai_unet_set_option, include/oai_hip.h); bit-preserving: sres, sres_mrep, sres_ring, "
                         "xcd_group, fuse_first, b_lds, wide, shared_enc, dead_stores, census; NOT bit-preserving: winograd (bit mask, default 19; "
                         "0 = direct form), winograd_layers, m16 (default 1: tap pairs in the direct kernel), m16_layers; repeatable")
    ap.add_argument("--no-overlap", action="store_true", help="registration after, not underneath, the segmentation (A/B of VolumePipeline.overlap_registration)")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo check of launcher + collectives, no GPU, no kernels")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ                                 # started by torch.distributed.run (also with one rank)
    if not launched and args.gpus > 1:
        if not args.dry_run:
            import torch                                                   # device_count() does not initialise the GPU on this image
            have = torch.cuda.device_count()
            if have < args.gpus:
                print(f"bench.py: --gpus {args.gpus} requested but this node exposes {have} GPU(s)", file=sys.stderr)
                raise SystemExit(2)
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE {world} != --gpus {args.gpus}", file=sys.stderr)
        raise SystemExit(2)
    if args.dry_run:
        return dry_run(args, world, rank)

    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    use_dist = launched
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # "nccl" is RCCL on ROCm

    from oai_analysis_2_amd import _lib
    from oai_analysis_2_amd.image import Image
    from oai_analysis_2_amd.parallel import tile_range_for_rank
    from oai_analysis_2_amd.pipeline import CROP_ZYX, OVERLAP_ZYX, TILE_ZYX, VolumePipeline
    from oai_analysis_2_amd.registration import IconEngine
    from oai_analysis_2_amd.segmentation.engine import UNetEngine, tile_grid
    from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume
    _lib.load()                                                           # fail loudly if the HIP library is missing

    unet_sd = make_unet_state_dict(0)
    icon_sd = make_icon_state_dict(0, last_scale=0.1)
    unet = UNetEngine(unet_sd, precision=args.precision)
    wino_f32 = True                                                       # the library's default (option "winograd_f32")
    for opt in args.option:
        name, _, value = opt.partition("=")
        unet.set_option(name, int(value))
        if name == "winograd_f32":
            wino_f32 = bool(int(value))
    icon = IconEngine(icon_sd)
    atlas = Image(make_volume(1000, VOL_SHAPE), [0.36, 0.36, 0.7], [0.0, 0.0, 0.0])
    pipe = VolumePipeline(unet, icon, atlas, batch=args.batch or None)
    pipe.overlap_registration = not args.no_overlap
    n_distinct = 2
    vols_np = [make_volume((100 * rank if args.mode == "replicas" else 0) + i, VOL_SHAPE) for i in range(n_distinct)]
    vols = [torch.from_numpy(v).cuda() for v in vols_np]                  # resident in HBM before timing
    meta = Image(vols_np[0], [0.36, 0.36, 0.7], [2.0, -3.0, 1.0])

    # fp16x3: the per-layer activation exponents are calibrated once per network, outside the timed region (two segmentation passes
    # over the first volume); rank 0's exponents go to every rank so that a tile-sharded volume is computed with one set
    calibration = None
    if args.precision == "fp16x3":
        passes = unet.calibrate_volume(vols[0], TILE_ZYX, OVERLAP_ZYX, CROP_ZYX, batch=args.batch or None) if rank == 0 or args.mode != "tileshard" else 0
        exps = torch.tensor(unet.act_exponents()[0], dtype=torch.int32, device="cuda")
        if use_dist and world > 1:
            dist.broadcast(exps, 0)
            unet.set_act_exponents(exps.tolist())
        calibration = {"passes": passes, "activation_exponents": exps.tolist(),
                       "note": "layer k stores x * 2^e[k] as fp16 term pairs: every layer's maximum in [2^10, 2^11) on the calibration volume; "
                               "a later volume outside [8, 65504] raises the range flag and is repeated in fp32"}

    def step(i):
        # check=False: the per-volume fp16 range flag is snapshotted on the device and read after the timed region (what
        # CohortRunner does at download time), so that no host synchronisation sits between volumes
        if args.mode == "tileshard":
            return pipe.run_sharded(vols[i % n_distinct] if rank == 0 else None, meta, check=False)   # the volume lives on rank 0: broadcast inside
        return pipe.run(vols[i % n_distinct], meta, check=False)

    if args.mode == "cohort":
        from oai_analysis_2_amd.parallel import VolumeQueue
        warm_q, timed_q = VolumeQueue(args.warmup * world), VolumeQueue(args.steps * world)      # constructed in the same order on every rank
        claimed = []

        def run_queue(q):
            # a rank claims its next volume when the one BEFORE the volume it has just queued is finished: the host runs one volume
            # ahead of its GPU (so the GPU never idles), not the whole queue ahead (kernel launches are asynchronous: unbounded, the
            # rank whose host thread is fastest would claim most of the cohort -- ADVICE r2)
            flags, events = [], []
            for i in q:
                claimed.append(i)
                flags.append(pipe.run(vols[i % n_distinct], meta, check=False).overflow)
                ev = torch.cuda.Event()
                ev.record()
                events.append(ev)
                if len(events) >= 2:
                    events[-2].synchronize()
            return flags

        run_queue(warm_q)
        torch.cuda.synchronize()
        unet.profile_read()
        unet.profile(True)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        claimed.clear()
        t0 = time.perf_counter()
        flags = run_queue(timed_q)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        conv_ms, conv_launches = unet.profile_read()
        unet.profile(False)
        overflow = any(int(f.item()) for f in flags if f is not None)
        my_volumes = len(claimed)
    else:
        dt, conv_ms, conv_launches, overflow = measure(step, unet, args.steps, args.warmup, use_dist, dist)
        my_volumes = args.steps
    streamed_rows = None
    if args.mode == "cohort" and not args.no_streamed:
        # (outside the timed region) every rank streams 12 volumes out of ITS host memory through CohortRunner at the same time: the first multi-GPU line
        # then holds the resident rate (`value`) AND the streamed rate with N processes' host copies and PCIe traffic on one host (VERDICT r5 #4c)
        if use_dist:
            dist.barrier()
        sf = streamed_from_host(pipe, n_volumes=12)
        mine_row = torch.tensor([sf["steady_state"]["ms_per_volume"] if sf["steady_state"] else 0.0, sf["value"],
                                 sf["host_memcpy"]["out_of_pinned_GBps"] or 0.0, sf["host_memcpy"]["into_pinned_GBps"] or 0.0], dtype=torch.float64, device="cuda")
        if use_dist:
            gathered = torch.zeros(world * 4, dtype=torch.float64, device="cuda")
            dist.all_gather_into_tensor(gathered, mine_row)
            streamed_rows = gathered.view(world, 4).tolist()
        else:
            streamed_rows = [mine_row.tolist()]
    rank_ms = None
    if use_dist:
        # every rank's own wall time of the timed region (the barrier-bracketed region is the same for all; what differs is how long each
        # rank's LAST step took to drain): gathered so that the line shows the spread, and the world size RCCL itself reports
        mine = torch.tensor([dt], dtype=torch.float64, device="cuda")
        every = torch.zeros(world, dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(every, mine)
        rank_ms = [1e3 * float(v) / max(my_volumes if args.mode == "cohort" else args.steps, 1) for v in every.tolist()]
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        _, _, n_tiles = tile_grid(VOL_SHAPE, TILE_ZYX, OVERLAP_ZYX)
        if args.mode in ("replicas", "cohort"):
            my_frac = 1.0
        else:                                        # rank 0's share of the volume's work under the cost-balanced tile split
            costs = unet.tile_costs(VOL_SHAPE, TILE_ZYX, OVERLAP_ZYX, CROP_ZYX)
            b, e = tile_range_for_rank(n_tiles, rank, world, costs)
            my_frac = sum(costs[b:e]) / sum(costs)
        # roofline.achieved uses SURVEY.md 8(d)'s contract figure: 505.4 GFLOP per tile = per-layer trim boxes of App. B.1
        # (the part of it that the 3x3x3 kernel runs).  The kernels additionally skip, per border tile, the part of the
        # kept centre that Partition.assemble zeroes (the 8/16/16 frame): that stricter "frame-aware" count is reported
        # beside it (achieved_frame_aware) so that nothing is overstated.
        survey_conv3 = unet.tile_flops_conv3(TILE_ZYX, OVERLAP_ZYX, True) * n_tiles
        vol_conv3 = unet.volume_flops(VOL_SHAPE, TILE_ZYX, OVERLAP_ZYX, CROP_ZYX, True, True)

        def traffic_of(prec):     # HBM-side bytes per launch: PMC counters need their own rocprofv3 passes, so this is READ FROM profiles/
            try:
                with open(os.path.join(ROOT, TRAFFIC_FILE[prec])) as f:
                    # the counters were collected on passes of `tiles_per_pass` tiles (160: this run's launch size); a launch here covers `last_batch`
                    js = json.load(f)
                    return js["bytes_per_launch"] * getattr(unet, "last_batch", 160) / float(js.get("tiles_per_pass", 32))
            except (OSError, KeyError, ValueError):
                return None

        def roofline(prec, ms, launches, steps, frac_of_volume):
            alg = survey_conv3 * frac_of_volume * steps
            ach = alg / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            ach_fa = ach * vol_conv3 / survey_conv3
            peak = MFMA_F32_PEAK_TFLOPS if prec == "f32" else MFMA_BF16_PEAK_TFLOPS
            return {"bound": "mfma", "kernel": KERNEL_OF[prec], "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                    "traffic": traffic_of(prec),
                    "traffic_source": (TRAFFIC_FILE.get(prec, "none") + " (rocprofv3 --pmc, separate passes of this round's library at this launch size; "
                                       "NOT measured in this run)") if traffic_of(prec) is not None else None,
                    "achieved_frame_aware": ach_fa, "frac_frame_aware": ach_fa / peak,
                    "mfma_passes_per_product": PASSES[prec],
                    "executed_frac": ach_fa * (WINO_F32_EXECUTED if prec == "f32" and wino_f32 else PASSES[prec]) / peak,
                    "executed_frac_of_sustained_issue_rate": None if prec == "f32" else ach_fa * PASSES[prec] / SUSTAINED_16BIT_MFMA_TFLOPS,
                    "executed_note": ("every 3x3x3 layer of the exact-fp32 path runs conv3_wino_f32: four GEMMs with K = 9 Cin instead of one with K = 27 Cin = 2/3 of the "
                                      "algorithmic MFMAs (unet_wino_f32.h); `frac` = algorithmic FLOP / time / peak can therefore exceed 1, executed_frac = frame-aware "
                                      "algorithmic x 2/3 / peak is the matrix pipe's load" if wino_f32 else None) if prec == "f32" else
                    "executed_frac counts 3 MFMA passes per ALGORITHMIC product; the k3 layers without a fused ec0 / head (ec3-ec7 dc8 dc7 dc5 dc4 dc2, 78 % of the "
                    "3x3x3 algorithmic FLOP) run the x axis in Winograd F(2,3) form and execute 2/3 of that (unet_wino.h, profiles/r03_winograd.md, r04_wino_stream.md): for them "
                    "it overstates the matrix pipe's load, `frac` (algorithmic FLOP / time / peak) is the contract figure",
                    "algorithmic_flops_per_launch": alg / max(launches, 1), "avg_launch_ms": ms / max(launches, 1), "launches": launches,
                    "clock_note": None if prec == "f32" else
                    "the 16-bit MFMA path is power-limited on this workload: sclk 1.96 GHz at ~1.28 kW (profiles/r01_power.md), i.e. a "
                    "dense peak of 2.05 PFLOP/s at the clock it runs at; `peak` stays the nominal 2.5 PFLOP/s",
                    "concurrency_note": "the ICON registration kernels of the same volume run on a side stream underneath these launches"}

        out = {
            "metric": "knee MRI volumes/sec (segment+register), 384x384x160 fp32",
            "value": (1 if args.mode == "tileshard" else world) * args.steps / dt, "unit": "volumes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong" if args.mode == "tileshard" else "weak", "vs_baseline": None,
            "dtype": DTYPE_OF[unet.effective_precision], "data": "synthetic",
            "config": {"workload": "fused segment->register->resample per volume, 1 volume per GPU per step, 384x384x160 fp32, "
                                   "160 tiles of 128x128x32 (overlap 16/16/8), ICON 80x192x192 one direction, FC+TC resample",
                       "tiles_per_pass": getattr(unet, "last_batch", args.batch), "parallelism": f"{args.mode} x{world}",
                       "collective_backend": ("nccl (RCCL)" if use_dist else None), "world_size": world,
                       "rccl_world_size": (dist.get_world_size() if use_dist else None),
                       "per_rank_ms_per_step": ({"min": min(rank_ms), "max": max(rank_ms)} if rank_ms else None),
                       "options": args.option or None,
                       "cohort": ({"volumes": args.steps * world, "claimed_by_rank0": my_volumes} if args.mode == "cohort" else None)},
            "roofline": roofline(unet.effective_precision, conv_ms, conv_launches, my_volumes, my_frac),
            "requested_precision": args.precision, "fp16_refused": unet.fp16_refused,
            "fp16_range_overflow": overflow,
            "fp16_calibration": calibration,
            "segment_algorithmic_tflop_per_volume": unet.tile_flops(TILE_ZYX, OVERLAP_ZYX, True) * n_tiles / 1e12,
            "segment_frame_aware_tflop_per_volume": unet.volume_flops(VOL_SHAPE, TILE_ZYX, OVERLAP_ZYX, CROP_ZYX, True, False) / 1e12,
        }
        if world == 1 and not args.no_parity:
            out["parity"] = fullsize_parity(unet, args.precision)
            if out["parity"] is not None:
                if args.precision == "fp16x3":
                    out["parity"]["rescaled_network"] = rescaled_network_parity(unet_sd)
                others = out["parity"]["other_cases"] = other_cases_parity(args.precision)
                # the headline arithmetic's own parity numbers in <= 200 bytes, inside `config`: flips and the larger
                # class's sum|dp| per 23.6 M voxels (reference budget 12) for base / bn / dc / win
                rows = [out["parity"]] + [others[k] for k in ("bn", "dc", "win") if k in others]
                out["config"]["parity_headline"] = {"precision": args.precision + ("+winograd" if args.precision == "fp16x3" and "winograd=0" not in args.option else ""),
                                                    "cases": ["base", "bn", "dc", "win"][:len(rows)],
                                                    "flips": [r["mask_flips"] for r in rows],
                                                    "sum_abs_dp": [round(max(r["sum_abs_dp_per_23.6M_voxels"]), 2) for r in rows],
                                                    # where the reference's ABSOLUTE acceptance budget (sum|dp| < 12, test/test_all.py:32-33) holds
                                                    "abs12": [bool(max(r["sum_abs_dp_per_23.6M_voxels"]) < 12.0) for r in rows],
                                                    "budget": "12 (base, win); max(12, 2x ref fp32 noise) (bn, dc)"}
        if world == 1 and not args.no_parity:
            A_net = torch.from_numpy(make_volume(1, (80, 192, 192))).cuda()
            B_net = torch.from_numpy(make_volume(2, (80, 192, 192))).cuda()
            out["registration_step_trees"] = icon_step_tree_times(A_net, B_net)
        if world == 1 and not args.no_streamed and args.mode == "replicas":
            out["streamed_from_host"] = streamed_from_host(pipe, resident_ms=1e3 * dt / args.steps)
        if streamed_rows is not None:
            out["streamed_from_host"] = aggregate_streamed(streamed_rows, resident_ms=1e3 * dt / args.steps)
        if world == 1 and not args.no_alt:
            # the SAME workload, same --steps / --warmup, with the other arithmetic (exact fp32 MFMA when the primary is split-fp16):
            # a first-class measurement, so that a reader who only credits reference-precision arithmetic has a number
            alt = "f32" if args.precision != "f32" else "fp16x3"
            unet.set_precision(alt)
            dta, ms_a, n_a, ov_a = measure(step, unet, args.steps, args.warmup, False, dist)
            unet.set_precision(args.precision)
            out["fp32_mfma" if alt == "f32" else "alt_precision"] = {
                "precision": alt, "dtype": DTYPE_OF[alt], "value": args.steps / dta, "unit": "volumes/s", "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": 1e3 * dta / args.steps, "roofline": roofline(alt, ms_a, n_a, args.steps, 1.0),
                "parity": None if args.no_parity else fullsize_parity(unet, alt)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(vols_np[0], meta, atlas, unet_sd, icon_sd)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
