"""Worker of tests/test_distributed_gpu.py::test_process_cohort_at_world_2_on_one_gpu_over_gloo: two ranks on device 0 (gloo), ONE volume queue, one
calibration for the group (rank 0 calibrates on volume 0, the store carries its outcome), a CalibrationBoard and a CohortRunner per rank
(dask_processing.process_cohort: the driver that replaces the reference's Dask graph, dask_processing.py:46-189).  Every rank reports the volumes it
claimed with a digest of their results; rank 0 also runs every volume on its own (one resident pipeline, no cohort machinery) and prints one JSON line.
Not a test module itself."""
import hashlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def digest(r):
    h = hashlib.sha256()
    for name in ("fc", "tc", "phi", "fc_atlas", "tc_atlas"):
        h.update(np.ascontiguousarray(getattr(r, name).cpu().numpy()).tobytes())
    return h.hexdigest()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    td = os.environ["OAI_TEST_DIR"]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from oai_analysis_2_amd import dask_processing as dp
    from oai_analysis_2_amd.image import Image
    from oai_analysis_2_amd.io_nifti import write_nifti
    from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume
    n_vol = 7
    if rank == 0:
        with open(os.path.join(td, "segmentation_train_config.pth.tar"), "w") as f:
            json.dump({"patch_size": [64, 64, 32], "model": "UNet", "model_setting": {"in_channels": 1, "n_classes": 2, "bias": True, "BN": False}}, f)
        torch.save({"model_state_dict": make_unet_state_dict(seed=4), "epoch": 3}, os.path.join(td, "segmentation_model.pth.tar"))
        for i in range(n_vol):
            write_nifti(os.path.join(td, f"knee{i}.nii.gz"), Image(make_volume(30 + i, (24, 72, 72)) * 900.0 + 17.0, [0.36, 0.37, 0.7], [1.0, 2.0, 3.0]))
    dist.barrier()
    paths = [os.path.join(td, f"knee{i}.nii.gz") for i in range(n_vol)]
    net = (40, 48, 48)
    dp.set_worker(dp.Worker(models_dir=td, icon_weights=make_icon_state_dict(3, last_scale=0.1), icon_net_shape=net))
    atlas = Image(make_volume(31, (40, 80, 88)), [0.4, 0.35, 0.75], [0.0, -1.0, 2.0])
    mine = {i: digest(r) for i, r in dp.process_cohort(paths, atlas)}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    exps = dp.get_worker().segmenter.model.engine.act_exponents()
    all_exps = [None] * world
    dist.all_gather_object(all_exps, exps)
    out = None
    if rank == 0:
        # the same volumes one by one through a plain resident pipeline on this rank (its engines hold the group's calibration)
        from oai_analysis_2_amd.pipeline import VolumePipeline
        w = dp.get_worker()
        seg = w.segmenter
        ovl = tuple(int(v) for v in seg.config["overlap_size"])
        pipe = VolumePipeline(seg.model.engine, w.registerer.register_module, atlas, tile_zyx=seg.tile_zyx, overlap_zyx=ovl[::-1], crop_zyx=(ovl[2], ovl[0], ovl[1]))
        alone = {}
        for i, p in enumerate(paths):
            img = dp.image_normalize(dp.readimage(p), 0.1, 99.9, 0, 1)
            alone[i] = digest(pipe.run(torch.from_numpy(np.ascontiguousarray(img.array, dtype=np.float32)).cuda(), img))
        out = {"world": world, "claimed": [sorted(m) for m in everyone], "same_calibration": all(e == all_exps[0] for e in all_exps), "calibrated": bool(all_exps[0][1]),
               "match_alone": {str(i): all(m.get(i, alone[i]) == alone[i] for m in everyone) for i in range(n_vol)}}
    dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
