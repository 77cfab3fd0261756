"""The one numeric helper of the reference's oai_analysis/dask_processing.py that sits on the path:
``image_normalize`` (:10-26).  The Dask task graph itself is replaced by :mod:`oai_analysis_2_amd.cohort`."""
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
