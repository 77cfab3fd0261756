"""GPU: the reference-surface classes (Segmenter3DInPatchClassWise, AnalysisObject, ICON_Registration,
deform_probmap) driven exactly like the reference's own tests drive them (test/test_all.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oai_analysis_2_amd.image import Image
from oai_analysis_2_amd.synth import make_icon_state_dict, make_unet_state_dict, make_volume
from oracle import icon as oicon, resample as oresample, seg as oseg

pytestmark = pytest.mark.gpu


def _write_models(td, patch, unet_seed, bn=False):
    with open(os.path.join(td, "segmentation_train_config.pth.tar"), "w") as f:     # JSON under a .pth.tar name
        json.dump({"patch_size": list(patch), "model": "UNet",
                   "model_setting": {"in_channels": 1, "n_classes": 2, "bias": True, "BN": bn}}, f)
    torch.save({"model_state_dict": make_unet_state_dict(seed=unet_seed, bn=bn), "epoch": 3, "best_score": 0.5},
               os.path.join(td, "segmentation_model.pth.tar"))


def test_segmenter_matches_reference_golden(golden_dir, tmp_path):
    from oai_analysis_2_amd.segmentation.segmenter import Segmenter3DInPatchClassWise
    z = np.load(os.path.join(golden_dir, "segment_small.npz"))
    _write_models(str(tmp_path), tuple(int(v) for v in z["patch"]), int(z["weight_seed"]))
    seg = Segmenter3DInPatchClassWise(mode="pred", config=dict(
        ckpoint_path=str(tmp_path / "segmentation_model.pth.tar"),
        training_config_file=str(tmp_path / "segmentation_train_config.pth.tar"),
        device="cuda", batch_size=4, overlap_size=tuple(int(v) for v in z["overlap"]), output_prob=True, output_itk=True))
    img = Image(make_volume(int(z["volume_seed"]), (24, 72, 72)), [0.36, 0.36, 0.7], [1, 2, 3])
    fc, tc = seg.segment(img, if_output_prob_map=True, if_output_itk=True)
    assert isinstance(fc, Image) and fc.array.dtype == np.float64 and np.allclose(fc.spacing, img.spacing)   # CopyInformation
    budget = 12.0 * img.array.size / 23592960
    assert np.abs(fc.array - z["fc_prob"]).sum() < budget and np.abs(tc.array - z["tc_prob"]).sum() < budget
    fm, tm = seg.segment(img, if_output_prob_map=False, if_output_itk=False)
    assert isinstance(fm, np.ndarray) and set(np.unique(fm)) <= {0.0, 1.0}
    assert (fm.astype(np.uint8) != z["fc_mask"]).sum() <= 3 and (tm.astype(np.uint8) != z["tc_mask"]).sum() <= 3


def test_segmenter_errors_like_the_reference(tmp_path):
    from oai_analysis_2_amd.segmentation.networks import get_network
    from oai_analysis_2_amd.segmentation.segmenter import Segmenter3DInPatchClassWise
    _write_models(str(tmp_path), (32, 32, 16), 1)
    cfg = dict(ckpoint_path=str(tmp_path / "missing.pth.tar"), training_config_file=str(tmp_path / "segmentation_train_config.pth.tar"),
               device="cuda", batch_size=4, overlap_size=(8, 8, 4), output_prob=True, output_itk=True)
    with pytest.raises(ValueError, match="no checkpoint found"):             # utils.py:41
        Segmenter3DInPatchClassWise(mode="pred", config=cfg).segment(Image(make_volume(0, (16, 32, 32))))
    cfg["ckpoint_path"] = str(tmp_path / "segmentation_model.pth.tar")
    cfg["device"] = "cpu"
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Segmenter3DInPatchClassWise(mode="pred", config=cfg).segment(Image(make_volume(0, (16, 32, 32))))
    assert get_network("nope") is None                                       # networks.py:858-862


def test_analysis_object_end_to_end(tmp_path):
    """test_all.py:35-58: register, then deform both probability maps onto the atlas."""
    from oai_analysis_2_amd.analysis_object import AnalysisObject
    from oai_analysis_2_amd.registration import DisplacementTransform, deform_probmap
    td = str(tmp_path)
    _write_models(td, (64, 64, 32), 4)
    icon_sd = make_icon_state_dict(3, last_scale=0.1)
    torch.save(icon_sd, os.path.join(td, "icon_weights.pth"))
    atlas = Image(make_volume(31, (40, 80, 88)), [0.4, 0.35, 0.75], [0.0, -1.0, 2.0])
    np.savez(os.path.join(td, "atlas_image.npz"), array=atlas.array, spacing=atlas.spacing, origin=atlas.origin, direction=atlas.direction)
    obj = AnalysisObject(models_dir=td)
    obj.registerer.register_module = type(obj.registerer.register_module)(icon_sd, net_shape=(40, 48, 48))   # small net grid
    img = Image(make_volume(30, (24, 72, 72)), [0.36, 0.37, 0.7], [1.0, 2.0, 3.0])
    FC, TC = obj.segment_volume(img)                                          # alias of .segment
    fc_ref, tc_ref = oseg.segment(img.array, make_unet_state_dict(4), (64, 64, 32), (16, 16, 8))
    assert np.abs(FC.array - fc_ref).max() < 1e-5 and np.abs(TC.array - tc_ref).max() < 1e-5
    phi = obj.register_to_atlas(img)                                          # alias of .register
    assert isinstance(phi, DisplacementTransform) and phi.displacement.shape == (40, 48, 48, 3)
    phi_ref, _ = oicon.register_pair_arrays(img.array, atlas.array, icon_sd, net_shape=(40, 48, 48), both=False)
    disp_ref = oicon.displacement_itk(phi_ref)
    assert np.abs(phi.displacement - disp_ref).max() < 1e-4 * max(1.0, np.abs(disp_ref).max())
    warped = deform_probmap(phi, img, obj.atlas_image, FC)
    ref = oresample.resample_through_phi(FC.array, disp_ref, img, atlas)
    assert warped.array.shape == atlas.array.shape and warped.array.dtype == np.float64
    assert np.abs(warped.array - ref).max() < 1e-4


def test_partition_call_and_assemble_like_the_reference():
    """Partition used directly (image_transforms.py:388-519): tiles, then assemble with the reference's crop_size indexing."""
    from oai_analysis_2_amd.segmentation.image_transforms import Partition
    vol = make_volume(4, (20, 50, 44))
    patch, ovl = (32, 24, 16), (6, 4, 3)                        # x,y,z ; x / y overlaps differ on purpose
    part = Partition(patch, ovl)
    tiles = part({"image": Image(vol, [0.5, 0.5, 1.0]), "name": "v"})["image"]
    ref_tiles, g = oseg.partition(vol, patch, ovl)
    assert np.array_equal(tiles.cpu().numpy()[:, 0], ref_tiles[:, 0] if ref_tiles.ndim == 5 else ref_tiles)
    for crop in (None, ovl, (6, 4, 0)):
        got = part.assemble(tiles[:, 0].cpu(), if_itk=False, crop_size=crop)
        ref = oseg.assemble(ref_tiles[:, 0] if ref_tiles.ndim == 5 else ref_tiles, g, crop_size_xyz=crop)
        assert got.dtype == np.float64 and np.array_equal(got, ref.astype(np.float64))
    img = part.assemble(tiles, if_itk=True, crop_size=ovl)
    assert isinstance(img, Image) and np.allclose(img.spacing, [0.5, 0.5, 1.0])
    with pytest.raises(IndexError):
        part.assemble(tiles, is_vote=True)                                    # float tiles cannot index the vote array (numpy semantics)


def test_partition_vote_assemble_matches_reference_golden(golden_dir):
    """Partition.assemble(is_vote=True) (image_transforms.py:466-484) against the reference's own output (partition_vote.npz):
    label voting over the overlaps, np.argmax tie rule, uint8 result, float64 once crop_size is applied; and the gather of
    Partition.__call__ (a HIP kernel) against the reference's tiles of the same cases."""
    from oai_analysis_2_amd.segmentation.image_transforms import Partition
    z = np.load(os.path.join(golden_dir, "partition_vote.npz"))
    for idx in (0, 1):
        patch, ovl = tuple(int(v) for v in z[f"v{idx}_patch"]), tuple(int(v) for v in z[f"v{idx}_overlap"])
        part = Partition(patch, ovl)
        tiles = part({"image": z[f"v{idx}_vol"], "name": ""})["image"]
        lab = torch.from_numpy(z[f"v{idx}_labels"].astype(np.int64))
        assert tuple(tiles.shape[2:]) == tuple(lab.shape[1:]) and tiles.shape[0] == lab.shape[0]
        got = part.assemble(lab, is_vote=True, if_itk=False, crop_size=None)
        assert got.dtype == np.uint8 and np.array_equal(got, z[f"v{idx}_vote"])
        got = part.assemble(lab, is_vote=True, if_itk=False, crop_size=ovl)
        assert got.dtype == np.float64 and np.array_equal(got, z[f"v{idx}_vote_crop"])
    with pytest.raises(IndexError):
        part.assemble(lab.float(), is_vote=True)                       # numpy: only integers are valid indices
    with pytest.raises(IndexError):
        part.assemble(lab * 5, is_vote=True)                           # label value >= number of label planes
    c = np.load(os.path.join(golden_dir, "partition_cases.npz"))      # __call__: the reference's tiles of the ragged cases
    for idx in (0, 1, 2):
        part = Partition(tuple(int(v) for v in c[f"c{idx}_patch"]), tuple(int(v) for v in c[f"c{idx}_overlap"]))
        tiles = part({"image": c[f"c{idx}_vol"], "name": ""})["image"]
        assert np.array_equal(tiles.cpu().numpy(), c[f"c{idx}_tiles"])


def test_dask_task_bodies_with_a_persistent_worker(tmp_path):
    """dask_processing.py:46-189 -- segment_method / register_images_delayed / deform_probmap_delayed keep their names, arguments and
    results, but run on ONE resident worker; process_cohort streams a cohort of files through it from a queue."""
    from oai_analysis_2_amd import dask_processing as dp
    from oai_analysis_2_amd.io_nifti import write_nifti
    from oai_analysis_2_amd.registration import DisplacementTransform
    td = str(tmp_path)
    _write_models(td, (64, 64, 32), 4)
    icon_sd = make_icon_state_dict(3, last_scale=0.1)
    net = (40, 48, 48)
    dp.set_worker(dp.Worker(models_dir=td, icon_weights=icon_sd, icon_net_shape=net))
    atlas = Image(make_volume(31, (40, 80, 88)), [0.4, 0.35, 0.75], [0.0, -1.0, 2.0])
    paths = []
    for i in range(3):
        p = os.path.join(td, f"knee{i}.nii.gz")
        write_nifti(p, Image(make_volume(30 + i, (24, 72, 72)) * 900.0 + 17.0, [0.36, 0.37, 0.7], [1.0, 2.0, 3.0]))   # raw intensities: the window matters
        paths.append(p)
    FC, TC = dp.segment_method(paths[0])
    img0 = dp.image_normalize(dp.readimage(paths[0]), 0.1, 99.9, 0, 1)
    assert 0.0 <= img0.array.min() and img0.array.max() <= 1.0
    fc_ref, tc_ref = oseg.segment(img0.array, make_unet_state_dict(4), (64, 64, 32), (16, 16, 8))
    assert np.abs(FC.array - fc_ref).max() < 1e-5 and np.abs(TC.array - tc_ref).max() < 1e-5
    phi, A, B = dp.register_images_delayed(paths[0], atlas)
    assert isinstance(phi, DisplacementTransform) and phi.displacement.shape == (*net, 3) and A.array.max() <= 1.0
    warped = dp.deform_probmap_delayed(phi, A, B, FC, image_type="FC")
    assert warped.array.shape == atlas.array.shape
    seen = {}
    for i, r in dp.process_cohort(paths, atlas):
        seen[i] = r
    assert sorted(seen) == [0, 1, 2]
    assert np.abs(seen[0].fc.numpy().astype(np.float64) - FC.array).max() < 1e-6                     # the same worker, the same arithmetic
    assert np.abs(seen[0].fc_atlas.numpy() - warped.array).max() < 1e-4
    dp.set_worker(None)
