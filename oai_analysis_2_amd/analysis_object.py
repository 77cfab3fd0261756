"""``AnalysisObject`` facade with the reference's surface (oai_analysis/analysis_object.py:9-49).

    obj = AnalysisObject()                 # assets from $OAI_DATA_DIR (the reference downloads them with pooch)
    FC, TC = obj.segment(image)            # float64 probability maps with the input's geometry
    phi    = obj.register(image)           # atlas-space -> patient-space transform

``segment_volume`` / ``register_to_atlas`` are aliases (names used by BASELINE.json's north star).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from .image import Image, as_image
from .registration import ICON_Registration
from .segmentation.segmenter import Segmenter3DInPatchClassWise


def data_dir() -> str:
    root = os.environ.get("OAI_DATA_DIR")
    if not root:
        raise ValueError("set OAI_DATA_DIR to a directory holding segmentation_model.pth.tar, "
                         "segmentation_train_config.pth.tar, icon_weights.pth and atlas_image.npz "
                         "(the reference fetches these with pooch, oai_analysis/data.py:8-22; no network here)")
    return root


def load_atlas(path: str) -> Image:
    """atlas_image.npz (array [z,y,x] + spacing/origin/direction) or a NIfTI file read by this package's own reader
    (io_nifti.read_nifti: ITK's LPS conventions; itk itself is not needed)."""
    if path.endswith(".npz"):
        z = np.load(path)
        return Image(z["array"], z["spacing"], z["origin"], z["direction"])
    from .io_nifti import read_nifti
    return read_nifti(path)


def asset_paths(root: str) -> dict:
    """Where the reference's three release tarballs (data.py:8-22, tag v2.0.0) put things when extracted under ``root`` --
    ``models/``, ``atlases/``, ``test_data/`` -- with a flat directory (everything next to each other) as the fallback."""
    models = os.path.join(root, "models") if os.path.isdir(os.path.join(root, "models")) else root
    atlas = os.path.join(root, "atlases", "atlas_60_LEFT_baseline_NMI", "atlas_image.nii.gz")     # analysis_object.py:40
    if not os.path.exists(atlas):
        atlas = os.path.join(models, "atlas_image.npz")
    return {"ckpoint_path": os.path.join(models, "segmentation_model.pth.tar"),
            "training_config_file": os.path.join(models, "segmentation_train_config.pth.tar"),
            "icon_weights": os.environ.get("OAI_ICON_WEIGHTS") or os.path.join(models, "icon_weights.pth"),
            "atlas": atlas, "test_case": os.path.join(root, "test_data", "colab_case")}


class AnalysisObject:
    def __init__(self, models_dir: Optional[str] = None, atlas_image=None, icon_weights=None, device: Optional[str] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: oai_analysis_2_amd is the MI355X path and has no CPU fallback")
        self.device = device or "cuda"
        paths = asset_paths(models_dir or data_dir())
        segmenter_config = dict(                                    # the literals of analysis_object.py:18-26
            ckpoint_path=paths["ckpoint_path"],
            training_config_file=paths["training_config_file"],
            device=self.device,
            batch_size=4,
            overlap_size=(16, 16, 8),
            output_prob=True,
            output_itk=True,
        )
        self.segmenter = Segmenter3DInPatchClassWise(mode="pred", config=segmenter_config)
        self.registerer = ICON_Registration(weights=icon_weights if icon_weights is not None else paths["icon_weights"], device=self.device)
        self.atlas_image = as_image(atlas_image) if atlas_image is not None else load_atlas(paths["atlas"])

    def segment(self, preprocessed_image):
        FC_probmap, TC_probmap = self.segmenter.segment(preprocessed_image, if_output_prob_map=True, if_output_itk=True)
        return (FC_probmap, TC_probmap)

    def register(self, preprocessed_image):
        return self.registerer.register(preprocessed_image, self.atlas_image)

    # aliases named in BASELINE.json (the reference itself has no such names)
    segment_volume = segment
    register_to_atlas = register
