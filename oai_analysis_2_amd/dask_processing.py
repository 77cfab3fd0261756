"""The task bodies of the reference's ``oai_analysis/dask_processing.py`` with PERSISTENT per-process workers.

The reference wraps these functions in ``dask.delayed`` and, inside every task, rebuilds its model
(``pretrained_models.OAI_knees_gradICON_model()`` at :77, ``Segmenter3DInPatchClassWise(...)`` at :170), moves it to the GPU, runs one
volume and deletes it again.  Here the same names take the same arguments and return the same things, but the engines behind them are
built ONCE per process (``get_worker``) and stay resident in HBM; ``process_cohort`` is the driver that replaces the task graph: every
rank pulls the next volume from one shared queue (``parallel.VolumeQueue``) and streams it through its GPU (``cohort.CohortRunner``:
upload of i+1, compute of i, download of i-1 overlap).

    image_normalize          :10-26    percentile window -> [0,1] (device kernel, oai_image_normalize)
    readimage                :29-43    itk.imread(path, itk.F)    (NIfTI via io_nifti)
    register_images_delayed  :46-92    read A, B; normalise A; ICON register_pair -> (phi_AB, image_A, image_B)
    deform_probmap_delayed   :95-111   ITK resample of a probability map through phi_AB onto image_B's grid
    get_thickness            :114-122  inner cartilage surface with per-point thickness
    segment_method           :125-189  read; normalise; Segmenter3DInPatchClassWise.segment -> (FC, TC) probability maps
"""
from __future__ import annotations

import os
from typing import Iterable, Iterator, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .image import Image, as_image


def image_normalize(image, window_min_perc, window_max_perc, output_min, output_max) -> Image:
    """Percentile intensity window -> [output_min, output_max], like the reference; computed on the GPU."""
    img = as_image(image)
    vol = torch.from_numpy(np.array(img.array, dtype=np.float32, order="C", copy=True)).cuda()      # (a file-backed array may be read-only)
    out = ops.image_normalize(vol, window_min_perc, window_max_perc, output_min, output_max)
    return img.like(out.cpu().numpy().astype(img.array.dtype if img.array.dtype.kind == "f" else np.float32))


def readimage(image_path) -> Image:
    """dask_processing.py:29-43: ``itk.imread(path, itk.F)`` (NIfTI here; see io_nifti.py).  Images pass through unchanged."""
    if isinstance(image_path, (Image, np.ndarray)):
        return as_image(image_path)
    from .io_nifti import read_nifti
    return read_nifti(str(image_path), np.float32)


class Worker:
    """What a Dask task of the reference rebuilds per call, built once per process: the segmenter and the ICON registration engine
    with their weights resident on the GPU.  ``models_dir`` = the reference's ``models/`` directory (or $OAI_DATA_DIR)."""

    def __init__(self, models_dir: Optional[str] = None, icon_weights=None, precision: str = "fp16x3", icon_net_shape=None):
        from .analysis_object import asset_paths, data_dir
        from .registration import ICON_Registration, NET_SHAPE
        from .segmentation.segmenter import Segmenter3DInPatchClassWise
        paths = asset_paths(models_dir or data_dir())
        self.segmenter = Segmenter3DInPatchClassWise(mode="pred", config=dict(      # the literals of dask_processing.py:160-168
            ckpoint_path=paths["ckpoint_path"], training_config_file=paths["training_config_file"], device="cuda",
            batch_size=2, overlap_size=(16, 16, 8), output_prob=True, output_itk=True, precision=precision))
        self.registerer = ICON_Registration(weights=icon_weights if icon_weights is not None else paths["icon_weights"],
                                            net_shape=icon_net_shape or NET_SHAPE, verbose=False)


_WORKER: Optional[Worker] = None


def get_worker(**kwargs) -> Worker:
    """The process's worker (created on first use; pass models_dir= / icon_weights= the first time, or set $OAI_DATA_DIR)."""
    global _WORKER
    if _WORKER is None:
        _WORKER = Worker(**kwargs)
    return _WORKER


def set_worker(worker: Optional[Worker]) -> None:
    global _WORKER
    _WORKER = worker


def segment_method(image_A):
    """dask_processing.py:125-189 without the per-task model download / construction / deletion."""
    test_volume = image_normalize(readimage(image_A), 0.1, 99.9, 0, 1)
    FC_probmap, TC_probmap = get_worker().segmenter.segment(test_volume, if_output_prob_map=True, if_output_itk=True)
    return FC_probmap, TC_probmap


def register_images_delayed(image_A, image_B):
    """dask_processing.py:46-92: (phi_AB, image_A (normalised), image_B)."""
    image_A, image_B = readimage(image_A), readimage(image_B)
    # the reference casts BOTH images to itk.D (:63-73) and windows A in double; here the window (percentiles + rescale) is computed in
    # fp32 on the device and widened: values agree with the double computation to ~1e-7 (tests/test_normalize_gpu.py)
    image_B = image_B.like(image_B.array.astype(np.float64))
    image_A = image_normalize(image_A.like(image_A.array.astype(np.float64)), 0.1, 99.9, 0, 1)
    phi_AB = get_worker().registerer.register(image_A, image_B)
    return phi_AB, image_A, image_B


def deform_probmap_delayed(phi_AB, image_A, image_B, prob, image_type="FC"):
    """dask_processing.py:95-111."""
    from .registration import deform_probmap
    return deform_probmap(phi_AB, image_A, image_B, prob)


def get_thickness(warped_image, mesh_type):
    """dask_processing.py:114-122: the inner surface of the cartilage with per-point thickness ("Distance").  The reference converts the
    vtk mesh with mp.get_itk_mesh only "to make it serializable" for Dask's pickling (mesh_processing.py:57-98); there is no task graph
    to pickle for here, so the mesh is returned as mesh_processing produces it."""
    from . import mesh_processing as mp
    distance_inner, _ = mp.get_thickness_mesh(warped_image, mesh_type=mesh_type)
    return distance_inner


def process_cohort(images: Sequence, atlas_image, worker: Optional[Worker] = None, keep_on_device: bool = False) -> Iterator[Tuple[int, object]]:
    """The driver that replaces the Dask graph of DaskComputation.ipynb for the dense path: every rank of the (optional) process group
    claims volumes from ONE queue and streams them through its GPU with the resident engines: normalise -> segment + register ->
    both probability maps on the atlas grid.  Yields (index, VolumeResult) for the volumes THIS rank processed.
    ``images``: paths or Images (all ranks pass the same list; only claimed entries are read)."""
    from .cohort import CohortRunner
    from .parallel import CalibrationBoard, VolumeQueue, sync_calibration
    from .pipeline import VolumePipeline
    w = worker or get_worker()
    seg = w.segmenter
    if not seg.ready:
        seg.pred_setup()
    eng = seg.model.engine
    if eng.precision != seg.config.get("precision", "fp16x3"):
        eng.set_precision(seg.config.get("precision", "fp16x3"))
    ovl = tuple(int(v) for v in seg.config["overlap_size"])
    if eng.precision == "fp16x3" and len(images):
        # ONE calibration for the cohort: the checkpoint's sidecar if there is one, else rank 0 calibrates on volume 0 (and writes the
        # sidecar); every rank takes rank 0's exponents, so a volume's maps do not depend on which rank the queue hands it to
        eng.set_calibration_file(seg.calibration_file, write=getattr(seg, "calibration_write", False))

        def _calibrate_on_first():
            vol0 = image_normalize(readimage(images[0]), 0.1, 99.9, 0, 1)
            eng.calibrate_volume(torch.from_numpy(np.ascontiguousarray(vol0.array, dtype=np.float32)).to(eng.device),
                                 seg.tile_zyx, ovl[::-1], (ovl[2], ovl[0], ovl[1]))
        sync_calibration(eng, _calibrate_on_first)
    pipe = VolumePipeline(eng, w.registerer.register_module, readimage(atlas_image), tile_zyx=seg.tile_zyx, overlap_zyx=ovl[::-1],
                          crop_zyx=(ovl[2], ovl[0], ovl[1]))
    import torch.distributed as dist
    board = CalibrationBoard() if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None
    runner = CohortRunner(pipe, keep_on_device=keep_on_device, board=board)

    class _Lazy(Sequence):                      # reads + normalises a volume when the runner asks for it (after it was claimed)
        def __len__(self_inner):
            return len(images)

        def __getitem__(self_inner, i):
            # on the runner's copy stream: the upload, the percentile kernels and the blocking .cpu() of image_normalize then wait for
            # earlier copies only, not for the volume whose compute was queued a moment ago (the overlap was lost)
            with torch.cuda.stream(runner.copy_stream):
                return image_normalize(readimage(images[i]), 0.1, 99.9, 0, 1)

    return runner.run(_Lazy(), queue=VolumeQueue(len(images)))
