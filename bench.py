#!/usr/bin/env python
"""bench.py -- knee MRI volumes/s (segment + register + FC/TC resample), 384x384x160 fp32.

One "step" = one synthetic DESS volume through the whole per-volume hot path on one GPU:
overlap-tiled 3D U-Net segmentation (160 tiles of 128x128x32, reference tiling) -> stitch -> ICON
registration to the atlas (one direction, what ICON_Registration.register returns) -> both probability
maps pulled onto the atlas grid through phi.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, volumes are independent units (the reference's own Dask model), every rank
processes its own volume per step, no data-path collective ("scaling": "weak").  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VOL_SHAPE = (160, 384, 384)
MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, exact fp32
MFMA_BF16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA
PASSES = {"f32": 1, "bf16x6": 6, "bf16x3": 3, "fp16x3": 3}
SUSTAINED_16BIT_MFMA_TFLOPS = 1857.0   # scripts/micro/mfma_peak.hip on this chip: operands in registers, every CU (profiles/r01_ablation.md)


def cpu_baseline(vol_np, meta_A, atlas_img, unet_sd, icon_sd, n_tiles_sample=4):
    """The oracle (CPU port of the reference algorithm) timed on this host, on a bounded sample."""
    from oai_analysis_2_amd.image import Image
    from oracle import icon as oicon, resample as oresample, seg as oseg
    threads = torch.get_num_threads()
    tiles, g = oseg.partition(vol_np, (128, 128, 32), (16, 16, 8))
    x = torch.from_numpy(np.ascontiguousarray(tiles[:n_tiles_sample]))
    oseg.unet_forward(x[:1], unet_sd)                                    # warm-up
    t0 = time.time()
    for i in range(n_tiles_sample):
        oseg.unet_forward(x[i:i + 1], unet_sd)
    t_tile = (time.time() - t0) / n_tiles_sample
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
    return {"value": 1.0 / t_vol, "unit": "volumes/s", "cores": threads, "kind": "port",
            "sample": f"{n_tiles_sample} of {g['n_tiles']} U-Net tiles (x{g['n_tiles'] / n_tiles_sample:.0f}), "
                      f"1 full ICON direction, {zs}/{atlas_img.array.shape[0]} slices of one resample (x{2 * atlas_img.array.shape[0] // zs}); "
                      f"s/tile={t_tile:.2f} s_register={t_reg:.1f} s_resample={t_res:.1f}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="tiles per U-Net pass (sizes the activation workspace); 0 = all that fit in 60%% of free HBM")
    ap.add_argument("--precision", default="fp16x3", choices=["fp16x3", "f32", "bf16x6", "bf16x3"],
                    help="arithmetic of the 3x3x3 conv layers.  fp16x3 (default): every fp32 operand split into two fp16 terms, "
                         "3 MFMA passes, fp32 accumulate -- fp32-grade results (same parity margins as f32 in tests/); "
                         "f32: exact fp32 MFMA; bf16x6 / bf16x3: split-bf16 with 6 / 3 passes")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "tileshard"],
                    help="N>1: replicas = one volume per rank per step (weak scaling, no collective); tileshard = every step "
                         "is ONE volume whose 160 tiles are split over the ranks + one RCCL all_gather (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra bf16x6 measurement reported beside the fp32 one")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    use_dist = "WORLD_SIZE" in os.environ          # launched by torch.distributed.run (also with one rank)
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
    icon = IconEngine(icon_sd)
    atlas = Image(make_volume(1000, VOL_SHAPE), [0.36, 0.36, 0.7], [0.0, 0.0, 0.0])
    pipe = VolumePipeline(unet, icon, atlas, batch=args.batch or None)
    n_distinct = 2
    vols_np = [make_volume((100 * rank if args.mode == "replicas" else 0) + i, VOL_SHAPE) for i in range(n_distinct)]
    vols = [torch.from_numpy(v).cuda() for v in vols_np]                  # resident in HBM before timing
    meta = Image(vols_np[0], [0.36, 0.36, 0.7], [2.0, -3.0, 1.0])

    def step(i):
        if args.mode == "tileshard":
            return pipe.run_sharded(vols[i % n_distinct], meta)
        return pipe.run(vols[i % n_distinct], meta)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    unet.profile_read()
    unet.profile(True)                                                    # HIP events around the dominant kernel
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = step(i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    conv_ms, conv_launches = unet.profile_read()
    unet.profile(False)
    overflow = unet.range_overflow() if args.precision == "fp16x3" else False
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        _, _, n_tiles = tile_grid(VOL_SHAPE, TILE_ZYX, OVERLAP_ZYX)
        if args.mode == "replicas":
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
        alg_conv3 = survey_conv3 * my_frac * args.steps
        achieved = alg_conv3 / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        achieved_fa = achieved * vol_conv3 / survey_conv3
        peak = MFMA_F32_PEAK_TFLOPS if args.precision == "f32" else MFMA_BF16_PEAK_TFLOPS
        def traffic_of(prec):     # HBM-side bytes per launch: PMC counters need their own rocprofv3 passes (profiles/)
            try:
                if prec not in ("f32", "fp16x3"):
                    return None
                with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json" if prec == "f32" else "r01_pmc_traffic_sres.json")) as f:
                    # the counters were collected on 32-tile passes; a launch of this run covers `last_batch` tiles
                    return json.load(f)["bytes_per_launch"] * getattr(unet, "last_batch", 32) / 32.0
            except (OSError, KeyError, ValueError):
                return None
        kernel_of = {"f32": "conv3_igemm_f32", "fp16x3": "conv3_igemm_sres (split-resident fp16x3)",
                     "bf16x3": "conv3_igemm_bf16s (split 16-bit)", "bf16x6": "conv3_igemm_bf16s (split 16-bit)"}
        clock_note = ("the 16-bit MFMA path is power-limited on this workload: sclk 1.96 GHz at ~1.28 kW (profiles/r01_power.md), "
                      "i.e. a dense peak of 2.05 PFLOP/s at the clock it runs at; `peak` stays the nominal 2.5 PFLOP/s")
        out = {
            "metric": "knee MRI volumes/sec (segment+register), 384x384x160 fp32",
            "value": (world if args.mode == "replicas" else 1) * args.steps / dt, "unit": "volumes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak" if args.mode == "replicas" else "strong", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x6": "bf16x6 (fp32 operands split into 3 bf16 terms, 6 MFMA passes, fp32 accumulate)",
                      "bf16x3": "bf16x3 (2 bf16 terms, 3 MFMA passes, fp32 accumulate)",
                      "fp16x3": "fp16x3 (every fp32 value held as 2 fp16 terms = 22 mantissa bits, 3 MFMA passes per product, fp32 accumulate; "
                                "fp32 in and out of every entry point)"}[args.precision], "data": "synthetic",
            "config": {"workload": "fused segment->register->resample per volume, 1 volume per GPU per step, 384x384x160 fp32, "
                                   "160 tiles of 128x128x32 (overlap 16/16/8), ICON 80x192x192 one direction, FC+TC resample",
                       "tiles_per_pass": getattr(unet, "last_batch", args.batch), "parallelism": f"{args.mode} x{world}"},
            "roofline": {"bound": "mfma", "kernel": kernel_of[args.precision],
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic_of(args.precision),
                         "clock_note": None if args.precision == "f32" else clock_note,
                         "concurrency_note": "the ICON registration kernels of the same volume run on a side stream underneath these "
                                             "launches (OAI_OVERLAP_REG=0 serialises: +1.5 % on volumes/s, the conv launches measure ~2 % longer)",
                         "achieved_frame_aware": achieved_fa, "frac_frame_aware": achieved_fa / peak,
                         "mfma_passes_per_product": PASSES[args.precision],
                         "executed_frac": achieved_fa * PASSES[args.precision] / peak,
                         "executed_frac_of_sustained_issue_rate": None if args.precision == "f32" else
                         achieved_fa * PASSES[args.precision] / SUSTAINED_16BIT_MFMA_TFLOPS,
                         "algorithmic_flops_per_launch": alg_conv3 / max(conv_launches, 1),
                         "avg_launch_ms": conv_ms / max(conv_launches, 1), "launches": conv_launches},
            "fp16_range_overflow": overflow,
            "segment_algorithmic_tflop_per_volume": unet.tile_flops(TILE_ZYX, OVERLAP_ZYX, True) * n_tiles / 1e12,
            "segment_frame_aware_tflop_per_volume": unet.volume_flops(VOL_SHAPE, TILE_ZYX, OVERLAP_ZYX, CROP_ZYX, True, False) / 1e12,
        }
        if world == 1 and not args.no_alt:
            # the same workload with the other arithmetic (exact fp32 MFMA when the primary is split-fp16, and vice versa);
            # reported beside the primary number, never as `value`
            alt = "f32" if args.precision != "f32" else "fp16x3"
            unet.set_precision(alt)
            step(0)
            torch.cuda.synchronize()
            unet.profile_read()
            unet.profile(True)
            ta = time.perf_counter()
            n_alt = min(2, args.steps)
            for i in range(n_alt):
                step(i)
            torch.cuda.synchronize()
            dta = time.perf_counter() - ta
            ms_a, n_a = unet.profile_read()
            unet.profile(False)
            unet.set_precision(args.precision)
            ach = survey_conv3 * n_alt / (ms_a * 1e-3) / 1e12
            pk = MFMA_F32_PEAK_TFLOPS if alt == "f32" else MFMA_BF16_PEAK_TFLOPS
            out["alt_precision"] = {"precision": alt, "value": n_alt / dta, "unit": "volumes/s", "ms_per_step": 1e3 * dta / n_alt,
                                    "roofline": {"bound": "mfma", "kernel": kernel_of[alt],
                                                 "achieved": ach, "peak": pk, "unit": "TFLOP/s", "frac": ach / pk, "traffic": traffic_of(alt),
                                                 "achieved_frame_aware": ach * vol_conv3 / survey_conv3,
                                                 "mfma_passes_per_product": PASSES[alt],
                                                 "executed_frac": PASSES[alt] * ach * vol_conv3 / survey_conv3 / pk},
                                    "parity": "same gates as the primary mode (tests/test_unet_gpu.py: logits <= 1e-4 rel, "
                                              "sum|dp| < 12 per 23.6M voxels)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(vols_np[0], meta, atlas, unet_sd, icon_sd)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
