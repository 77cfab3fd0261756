"""Worker of tests/test_distributed_gpu.py: started by `python -m torch.distributed.run --nproc-per-node 1` (the GPU box has one MI355X).
Initialises the "nccl" (= RCCL) process group BEFORE anything else touches the GPU, then runs one small volume through
VolumePipeline.run and through run_sharded (broadcast, tile ranges, all_gather of blocks, range-state all-reduce, slab gather: every
collective of parallel.py on device tensors) and prints one JSON line.  Not a test module itself (no test_ functions)."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("OAI_TEST_BACKEND", "nccl")
    if backend == "gloo":
        # world > 1 on a ONE-GPU box: every rank on device 0, the collectives over gloo ON DEVICE TENSORS (staged through the host by the backend).  Not
        # RCCL -- two ranks cannot share a device there -- but the first configuration in which parallel.py's sharded path (ragged tile ranges written
        # straight into the gather buffer's slots, the in-place all_gather, the stitch's slot table, the range-state all-reduce, the slab gather) runs
        # between two real processes on GPU memory
        local = 0
        torch.cuda.set_device(0)
        dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from oai_analysis_2_amd import parallel
    from oai_analysis_2_amd.image import Image
    from oai_analysis_2_amd.pipeline import VolumePipeline
    from oai_analysis_2_amd.registration import IconEngine
    from oai_analysis_2_amd.segmentation.engine import UNetEngine
    from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume
    shape, net = (24, 72, 72), (40, 48, 48)
    atlas = Image(make_volume(10, shape), [0.4, 0.35, 0.75], [0.0, -1.0, 2.0])
    unet = UNetEngine(make_unet_state_dict(1, width_div=2), precision="fp16x3")
    pipe = VolumePipeline(unet, IconEngine(make_icon_state_dict(1, last_scale=0.1), net_shape=net), atlas,
                          tile_zyx=(16, 32, 32), overlap_zyx=(4, 8, 8), crop_zyx=(4, 8, 8), batch=8)
    vol = make_volume(9, shape)
    meta = Image(vol, [0.36, 0.37, 0.7], [1.0, 2.0, 3.0])
    v = torch.from_numpy(vol).cuda()
    # one calibration for the group, through the store (rank 0 calibrates)
    exps = parallel.sync_calibration(unet, lambda: unet.calibrate_volume(v, pipe.tile_zyx, pipe.overlap_zyx, pipe.crop_zyx))
    one = pipe.run(v, meta)
    sh = pipe.run_sharded(v if rank == 0 else None, meta)
    torch.cuda.synchronize()
    from oai_analysis_2_amd.segmentation.engine import tile_grid
    n_tiles = tile_grid(shape, pipe.tile_zyx, pipe.overlap_zyx)[2]
    ranges = [list(parallel.tile_range_for_rank(n_tiles, r, world, unet.tile_costs(shape, pipe.tile_zyx, pipe.overlap_zyx, pipe.crop_zyx))) for r in range(world)]
    out = {"world": world, "backend": dist.get_backend(), "calibrated": unet.act_exponents()[1], "exponents": exps, "tile_ranges": ranges, "n_tiles": n_tiles,
           "equal": bool(all(torch.equal(getattr(one, k), getattr(sh, k)) for k in ("fc", "tc", "phi", "fc_atlas", "tc_atlas"))),
           "flag": int(sh.overflow.item()) if sh.overflow is not None else None, "fc_sum": float(sh.fc.double().sum())}
    ok = torch.tensor([1 if out["equal"] else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)                        # every rank's run_sharded must equal its own run
    out["equal_on_every_rank"] = bool(int(ok.item()))
    dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
