"""MI355X-native (gfx950) implementation of OAI_analysis_2's per-volume dense path.

Public surface mirrors the reference: ``AnalysisObject`` (analysis_object.py), ``segmentation.segmenter.
Segmenter3DInPatchClassWise``, ``registration.ICON_Registration``.  All compute runs in hand-written HIP
kernels behind the C ABI of ``liboai_hip.so`` (include/oai_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"


def imread(path, dtype="float32"):
    """``itk.imread(path, itk.F)`` without ITK: NIfTI-1 (.nii / .nii.gz) -> ``image.Image`` in ITK's LPS conventions."""
    import numpy as _np
    from .io_nifti import read_nifti
    return read_nifti(path, _np.dtype(dtype) if dtype is not None else None)


def imwrite(image, path):
    """``itk.imwrite(image, path)`` without ITK (NIfTI-1; gzip when the name ends in .gz)."""
    from .image import as_image
    from .io_nifti import write_nifti
    write_nifti(path, as_image(image))
