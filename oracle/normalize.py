"""Oracle: intensity windowing before the hot path (CPU, numpy).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates ``image_normalize`` (oai_analysis/dask_processing.py:10-26): the percentiles are numpy's own
``np.percentile`` (exactly what the reference calls); the windowing is ITK's IntensityWindowingImageFilter
functor for an ``itk.F`` image (ITK is absent here, PARITY UNPINNED for that half): window bounds cast to the pixel
type, factor/offset in double (NumericTraits<float>::RealType), result cast back to float.
"""
import numpy as np


def image_normalize(arr: np.ndarray, window_min_perc=0.1, window_max_perc=99.9, output_min=0.0, output_max=1.0):
    arr = np.asarray(arr, dtype=np.float32)
    wmin = np.float32(np.percentile(arr, window_min_perc))
    wmax = np.float32(np.percentile(arr, window_max_perc))
    factor = (np.float64(output_max) - np.float64(output_min)) / (np.float64(wmax) - np.float64(wmin))
    offset = np.float64(output_min) - np.float64(wmin) * factor
    out = (arr.astype(np.float64) * factor + offset).astype(np.float32)
    out[arr < wmin] = np.float32(output_min)
    out[arr > wmax] = np.float32(output_max)
    return out, (wmin, wmax)
