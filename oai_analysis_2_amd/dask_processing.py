"""The numeric task bodies of the reference's oai_analysis/dask_processing.py without the ``@delayed`` wrappers:
``image_normalize`` (:10-26), ``readimage`` (:29-43), ``get_thickness`` (:114-122).  Segmentation, registration and the
prob-map deformation are :mod:`oai_analysis_2_amd.analysis_object` / :mod:`.registration`; the Dask task graph itself is
replaced by :mod:`oai_analysis_2_amd.cohort`."""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .image import Image, as_image


def image_normalize(image, window_min_perc, window_max_perc, output_min, output_max) -> Image:
    """Percentile intensity window -> [output_min, output_max], like the reference; computed on the GPU."""
    img = as_image(image)
    vol = torch.from_numpy(np.ascontiguousarray(img.array, dtype=np.float32)).cuda()
    out = ops.image_normalize(vol, window_min_perc, window_max_perc, output_min, output_max)
    return img.like(out.cpu().numpy().astype(img.array.dtype if img.array.dtype.kind == "f" else np.float32))


def readimage(image_path) -> Image:
    """dask_processing.py:29-43: ``itk.imread(path, itk.F)`` (NIfTI here; see io_nifti.py)."""
    from .io_nifti import read_nifti
    return read_nifti(str(image_path), np.float32)


def get_thickness(warped_image, mesh_type):
    """dask_processing.py:114-122: the inner surface of the cartilage with per-point thickness ("Distance")."""
    from . import mesh_processing as mp
    distance_inner, _ = mp.get_thickness_mesh(warped_image, mesh_type=mesh_type)
    return distance_inner
