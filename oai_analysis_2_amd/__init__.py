"""MI355X-native (gfx950) implementation of OAI_analysis_2's per-volume dense path.

Public surface mirrors the reference: ``AnalysisObject`` (analysis_object.py), ``segmentation.segmenter.
Segmenter3DInPatchClassWise``, ``registration.ICON_Registration``.  All compute runs in hand-written HIP
kernels behind the C ABI of ``liboai_hip.so`` (include/oai_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
