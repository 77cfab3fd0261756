"""BASELINE config 1 -- the bundled test knee through ``AnalysisObject`` -- the day the assets are supplied.

The reference's only numeric pin is test/test_all.py:17-33: sum|FC - FC_probmap.nii.gz| < 12 and the same for TC, on
``colab_case/image_preprocessed.nii.gz`` with the released weights.  All of it is pooch downloads (oai_analysis/data.py:8-22,
release v2.0.0: oai-analysis-test-data / -atlases / -models .tar.gz) and there is no network here, so these tests SKIP WITH A
REASON unless ``OAI_DATA_DIR`` points at a directory with the three tarballs extracted (``models/``, ``atlases/``,
``test_data/``).  ICON weights come from icon_registration's own download: ``$OAI_ICON_WEIGHTS`` or ``models/icon_weights.pth``."""
import os

import numpy as np
import pytest

ROOT = os.environ.get("OAI_DATA_DIR", "")


def _paths():
    from oai_analysis_2_amd.analysis_object import asset_paths
    return asset_paths(ROOT)


def _missing(keys):
    if not ROOT:
        return ["$OAI_DATA_DIR is not set"]
    p = _paths()
    need = [p[k] for k in keys] + [os.path.join(p["test_case"], f) for f in ("image_preprocessed.nii.gz", "FC_probmap.nii.gz", "TC_probmap.nii.gz")]
    return [f for f in need if not os.path.exists(f)]


REASON = ("real OAI assets absent (pooch downloads of oai_analysis/data.py:8-22, no network in this environment): "
          "set OAI_DATA_DIR to the extracted v2.0.0 tarballs; missing: %s")
pytestmark = pytest.mark.gpu


@pytest.mark.skipif(bool(_missing(["ckpoint_path", "training_config_file"])), reason=REASON % _missing(["ckpoint_path", "training_config_file"])[:3])
@pytest.mark.parametrize("precision", ["fp16x3", "f32"])
def test_segmentation_of_the_bundled_knee_matches_the_released_probmaps(precision):
    """test/test_all.py:17-33 verbatim, through io_nifti.read_nifti + Segmenter3DInPatchClassWise with AnalysisObject's literals."""
    from oai_analysis_2_amd.io_nifti import read_nifti
    from oai_analysis_2_amd.segmentation.segmenter import Segmenter3DInPatchClassWise
    p = _paths()
    img = read_nifti(os.path.join(p["test_case"], "image_preprocessed.nii.gz"))
    seg = Segmenter3DInPatchClassWise(mode="pred", config=dict(                       # analysis_object.py:18-26
        ckpoint_path=p["ckpoint_path"], training_config_file=p["training_config_file"], device="cuda", batch_size=4,
        overlap_size=(16, 16, 8), output_prob=True, output_itk=True, precision=precision))
    FC, TC = seg.segment(img, if_output_prob_map=True, if_output_itk=True)
    for got, name in ((FC, "FC_probmap.nii.gz"), (TC, "TC_probmap.nii.gz")):
        ref = read_nifti(os.path.join(p["test_case"], name), dtype=np.float64)
        assert got.array.shape == ref.array.shape
        diff = np.abs(got.array - ref.array).sum()                                   # itk.comparison_image_filter(...).sum()
        print(f"[real knee {precision}] sum|d| vs {name}: {diff:.4f} (reference accepts < 12)")
        assert diff < 12
    # label maps bit-exact after thresholding (north star), except voxels the released map itself puts within 1e-5 of 0.5
    fm, tm = seg.segment(img, if_output_prob_map=False, if_output_itk=False)
    for got, name in ((fm, "FC_probmap.nii.gz"), (tm, "TC_probmap.nii.gz")):
        ref = read_nifti(os.path.join(p["test_case"], name), dtype=np.float64).array
        flips = (got > 0.5) != (ref > 0.5)
        print(f"[real knee {precision}] mask flips vs {name}: {int(flips.sum())}")
        assert np.all(np.abs(ref[flips] - 0.5) < 1e-5)


@pytest.mark.skipif(bool(_missing(["ckpoint_path", "training_config_file", "icon_weights", "atlas"])),
                    reason=REASON % _missing(["ckpoint_path", "training_config_file", "icon_weights", "atlas"])[:3])
def test_analysis_object_registers_and_deforms_the_bundled_knee():
    """test/test_all.py:35-58 up to the mesh step: AnalysisObject() from the assets, register, deform both released maps."""
    from oai_analysis_2_amd.analysis_object import AnalysisObject
    from oai_analysis_2_amd.io_nifti import read_nifti
    from oai_analysis_2_amd.registration import deform_probmap
    p = _paths()
    obj = AnalysisObject(models_dir=ROOT)
    img = read_nifti(os.path.join(p["test_case"], "image_preprocessed.nii.gz"))
    phi = obj.register(img)
    assert phi.displacement.shape == (80, 192, 192, 3) and np.isfinite(phi.displacement).all()
    for name in ("FC_probmap.nii.gz", "TC_probmap.nii.gz"):
        warped = deform_probmap(phi, img, obj.atlas_image, read_nifti(os.path.join(p["test_case"], name), dtype=np.float64))
        assert warped.array.shape == obj.atlas_image.array.shape and 0.0 <= warped.array.min() and warped.array.max() <= 1.0 + 1e-6
        assert warped.array.sum() > 0                                              # cartilage landed inside the atlas grid


# ---- the same code path on a stand-in asset tree: pins nothing new, proves that day one executes -----------------------------

@pytest.fixture(scope="module")
def standin_tree(tmp_path_factory, golden_dir):
    from oai_analysis_2_amd.synth import write_standin_asset_tree
    root = str(tmp_path_factory.mktemp("standin_assets"))
    info = write_standin_asset_tree(root, os.path.join(golden_dir, "segment_small.npz"))
    return root, info


@pytest.mark.parametrize("precision", ["fp16x3", "f32"])
def test_the_day_one_path_executes_on_a_standin_asset_tree(standin_tree, precision):
    """test/test_all.py:17-33's assertions through the SAME code as the real-asset test above -- the release layout under a data directory,
    ``asset_paths``, ``initialize_model`` on a ``torch.save({"model_state_dict": ...})`` checkpoint, the JSON training config, NIfTI in, the
    segmenter with AnalysisObject's literals, sum|d| against stored NIfTI maps -- on a tree written by oai_analysis_2_amd.synth.
    write_standin_asset_tree: the seeded 24 x 72 x 72 volume and the REFERENCE's own FC / TC maps for it (tests/golden/segment_small.npz).
    The only literal that differs from analysis_object.py:23 is overlap_size: the fixture was made with (8, 8, 4) on 32 x 32 x 16 patches."""
    from oai_analysis_2_amd.analysis_object import asset_paths
    from oai_analysis_2_amd.io_nifti import read_nifti
    from oai_analysis_2_amd.segmentation.segmenter import Segmenter3DInPatchClassWise
    root, info = standin_tree
    p = asset_paths(root)
    assert all(os.path.isfile(p[k]) for k in ("ckpoint_path", "training_config_file", "icon_weights", "atlas"))
    img = read_nifti(os.path.join(p["test_case"], "image_preprocessed.nii.gz"))
    seg = Segmenter3DInPatchClassWise(mode="pred", config=dict(
        ckpoint_path=p["ckpoint_path"], training_config_file=p["training_config_file"], device="cuda", batch_size=4,
        overlap_size=tuple(info["overlap_size"]), output_prob=True, output_itk=True, precision=precision))
    FC, TC = seg.segment(img, if_output_prob_map=True, if_output_itk=True)
    budget = 12.0 * img.array.size / 23592960                                       # the reference's 12 is per 384 x 384 x 160 voxels
    for got, name in ((FC, "FC_probmap.nii.gz"), (TC, "TC_probmap.nii.gz")):
        ref = read_nifti(os.path.join(p["test_case"], name), dtype=np.float64)
        assert got.array.shape == ref.array.shape and got.array.dtype == np.float64
        assert np.allclose(got.spacing, ref.spacing) and np.allclose(got.origin, ref.origin)
        diff = np.abs(got.array - ref.array).sum()
        print(f"[stand-in knee {precision}] sum|d| vs {name}: {diff:.6f} (< 12 as the reference asserts; scaled budget {budget:.4f})")
        assert diff < 12 and diff < budget
    fm, tm = seg.segment(img, if_output_prob_map=False, if_output_itk=False)
    for got, name in ((fm, "FC_probmap.nii.gz"), (tm, "TC_probmap.nii.gz")):
        ref = read_nifti(os.path.join(p["test_case"], name), dtype=np.float64).array
        flips = (got > 0.5) != (ref > 0.5)
        assert np.all(np.abs(ref[flips] - 0.5) < 1e-5)
    assert not os.path.exists(p["ckpoint_path"] + ".fp16cal.json")                  # (no side-effect file next to the checkpoint: ADVICE r4)


def test_analysis_object_runs_from_a_standin_asset_tree(standin_tree, monkeypatch):
    """test/test_all.py:35-58 up to the mesh step: ``AnalysisObject()`` with NO arguments from $OAI_DATA_DIR (the reference's pooch cache), the
    atlas read from the release layout's NIfTI, ``register`` + ``deform_probmap`` of both stored maps, ``segment`` through the facade."""
    from oai_analysis_2_amd.analysis_object import AnalysisObject, asset_paths
    from oai_analysis_2_amd.io_nifti import read_nifti
    from oai_analysis_2_amd.registration import deform_probmap
    root, info = standin_tree
    monkeypatch.setenv("OAI_DATA_DIR", root)
    obj = AnalysisObject()
    p = asset_paths(root)
    img = read_nifti(os.path.join(p["test_case"], "image_preprocessed.nii.gz"))
    assert obj.atlas_image.array.shape == tuple(info["shape_zyx"])
    phi = obj.register(img)
    assert phi.displacement.shape == (80, 192, 192, 3) and np.isfinite(phi.displacement).all()
    for name in ("FC_probmap.nii.gz", "TC_probmap.nii.gz"):
        warped = deform_probmap(phi, img, obj.atlas_image, read_nifti(os.path.join(p["test_case"], name), dtype=np.float64))
        assert warped.array.shape == obj.atlas_image.array.shape and 0.0 <= warped.array.min() and warped.array.max() <= 1.0 + 1e-6
        assert warped.array.sum() > 0
    # the facade's segment: AnalysisObject fixes overlap_size = (16, 16, 8) (analysis_object.py:23), which the stand-in's 32 x 32 x 16 patches cannot
    # hold -- the reference's error path for an impossible geometry, through the facade
    with pytest.raises(ValueError, match="overlap"):
        obj.segment(img)
    obj.segmenter.config["overlap_size"] = tuple(info["overlap_size"])
    FC, TC = obj.segment_volume(img)                                                # (the alias BASELINE.json names)
    ref = read_nifti(os.path.join(p["test_case"], "FC_probmap.nii.gz"), dtype=np.float64)
    assert np.abs(FC.array - ref.array).sum() < 12 and TC.array.shape == ref.array.shape
