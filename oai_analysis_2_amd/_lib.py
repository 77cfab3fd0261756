"""ctypes binding of liboai_hip.so (include/oai_hip.h).  Fails loudly: there is no fallback."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OAI_LIB_PATH") or os.path.join(HERE, "liboai_hip.so")   # override: diagnostic builds only


class OaiError(RuntimeError):
    pass


class Affine(C.Structure):
    _fields_ = [("A", C.c_double * 9), ("b", C.c_double * 3)]


class LayerParams(C.Structure):
    _fields_ = [("kind", C.c_int), ("cin", C.c_int), ("cout", C.c_int),
                ("weight", C.c_void_p), ("bias", C.c_void_p),
                ("bn_gamma", C.c_void_p), ("bn_beta", C.c_void_p), ("bn_mean", C.c_void_p), ("bn_var", C.c_void_p)]


class IconUnetParams(C.Structure):
    _fields_ = [("down_w", C.c_void_p * 5), ("down_b", C.c_void_p * 5),
                ("up_w", C.c_void_p * 5), ("up_b", C.c_void_p * 5),
                ("bn_gamma", C.c_void_p * 5), ("bn_beta", C.c_void_p * 5),
                ("bn_mean", C.c_void_p * 5), ("bn_var", C.c_void_p * 5),
                ("last_w", C.c_void_p), ("last_b", C.c_void_p)]


class IconNode(C.Structure):              # == struct oai_icon_node
    _fields_ = [("kind", C.c_int), ("a", C.c_int), ("b", C.c_int)]


# name -> (restype, argtypes); every symbol declared in include/oai_hip.h
_I, _P, _F, _Z, _D = C.c_int, C.c_void_p, C.c_float, C.c_size_t, C.c_double
_I3 = C.POINTER(C.c_int)
SIGNATURES = {
    "oai_version": (_I, []),
    "oai_last_error": (C.c_char_p, []),
    "oai_device_info": (_I, [C.c_char_p, _I]),
    "oai_grid_sample3d": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P]),
    "oai_compose": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _P, _P]),
    "oai_avgpool2_3d": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "oai_resize_trilinear": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P]),
    "oai_phi_to_itk_displacement": (_I, [_P, _I, _I, _I, _P, _P]),
    "oai_resample_through_disp": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, C.POINTER(Affine), C.POINTER(Affine),
                                       _P, _I, _I, _I, _P]),
    "oai_warp_chain": (_I, [_P, _I, _I, _I, _I, C.POINTER(_P), C.POINTER(_I), _P, _I, _I, _I, _P, _P]),
    "oai_resample_maps_through_phi": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, C.POINTER(Affine), C.POINTER(Affine),
                                           _P, _I, _I, _I, _P]),
    "oai_unet_tile_costs": (_I, [_P, _I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), _I]),
    "oai_mc_table": (_I, [_P]),
    "oai_mc_workspace_bytes": (_Z, [_I, _I, _I]),
    "oai_mc_count": (_I, [_P, _I, _I, _I, _F, _P, _Z, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), _P]),
    "oai_mc_emit": (_I, [_P, _I, _I, _I, _F, C.POINTER(C.c_float), _P, _P, _P, _P]),
    "oai_mesh_smooth": (_I, [_P, C.c_longlong, _P, _P, _I, _F, _P, _P, _P]),
    "oai_mesh_point_distance": (_I, [_P, C.c_longlong, _P, _P, C.c_longlong, _P, _P]),
    "oai_mesh_grid_workspace_bytes": (_Z, [C.POINTER(C.c_int), C.c_longlong]),
    "oai_mesh_point_distance_grid": (_I, [_P, C.c_longlong, _P, _P, C.c_longlong, C.POINTER(C.c_float), _F, C.POINTER(C.c_int), _P, _Z, _P, _P]),
    "oai_image_normalize_workspace_bytes": (_Z, []),
    "oai_image_normalize": (_I, [_P, _Z, _F, _F, _F, _F, _P, _P, _P, _Z, _P]),
    "oai_partition_tiles": (_I, [_P, _I, _I, _I, _I3, _I3, _I, _I, _P, _P]),
    "oai_assemble_vote": (_I, [_P, _I, _I, _I, _I, _I3, _I3, _P, _P]),
    "oai_unet_create": (_I, [C.POINTER(LayerParams), _F, C.POINTER(_P)]),
    "oai_unet_destroy": (None, [_P]),
    "oai_unet_set_precision": (_I, [_P, _I]),
    "oai_unet_range_flag": (_I, [_P, _I, C.POINTER(_I), _P]),
    "oai_unet_range_flag_snapshot": (_I, [_P, _P, _P]),
    "oai_unet_range_state_snapshot": (_I, [_P, _P, _P]),
    "oai_unet_range_flag_from_state": (_I, [_P, _P, _P]),
    "oai_unet_census": (_I, [_P, C.POINTER(_F), _I, _P]),
    "oai_unet_calibrate_step": (_I, [_P, _P, C.POINTER(_I)]),
    "oai_unet_get_act_exponents": (_I, [_P, C.POINTER(_I), C.POINTER(_I)]),
    "oai_unet_set_act_exponents": (_I, [_P, C.POINTER(_I)]),
    "oai_unet_set_option": (_I, [_P, C.c_char_p, _I]),
    "oai_unet_workspace_bytes": (_Z, [_P, _I, _I, _I, _I]),
    "oai_segment_workspace_bytes": (_Z, [_P, _I, _I, _I, _I3, _I3, _I]),
    "oai_unet_forward_tiles": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _Z, _P]),
    "oai_segment_tiles": (_I, [_P, _P, _I, _I, _I, _I3, _I3, _I3, _I, _I, _I, _P, _I, _P, _Z, _P]),
    "oai_unet_volume_flops": (_D, [_P, _I, _I, _I, _I3, _I3, _I3, _I, _I]),
    "oai_warp_set_option": (_I, [C.c_char_p, _I]),
    "oai_stitch_blocks": (_I, [_P, _I, _I, _I, _I, _I3, _I3, _I3, _P, _P]),
    "oai_stitch_blocks_ranged": (_I, [_P, _I, _I, _I, _I, _I3, _I3, _I3, _P, _I, _I, _P, _P]),
    "oai_unet_tile_flops": (_D, [_P, _I, _I, _I, _I3, _I]),
    "oai_unet_tile_flops_conv3": (_D, [_P, _I, _I, _I, _I3, _I]),
    "oai_unet_profile": (_I, [_P, _I]),
    "oai_unet_profile_read": (_I, [_P, C.POINTER(_D), C.POINTER(C.c_longlong)]),
    "oai_icon_create": (_I, [C.POINTER(IconUnetParams), _I, C.POINTER(IconNode), _I, _I, _I, _I, _I, C.POINTER(_P)]),
    "oai_icon_describe": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "oai_icon_destroy": (None, [_P]),
    "oai_icon_workspace_bytes": (_Z, [_P]),
    "oai_icon_forward": (_I, [_P, _P, _P, _P, _P, _Z, _P]),
    "oai_icon_set_graph": (_I, [_P, _I]),
    "oai_icon_set_option": (_I, [_P, C.c_char_p, _I]),
    "oai_icon_graph_info": (_I, [_P, C.POINTER(_I), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "oai_icon_unet_forward": (_I, [_P, _I, _P, _P, _I, _I, _I, _P, _P, _Z, _P]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load liboai_hip.so and bind every declared symbol; raise if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64; import it first so that liboai_hip.so binds to the SAME HIP
    # runtime instance (two runtimes in one process do not share devices, streams or allocations).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise OaiError(f"{LIB_PATH} is missing: build it with `python -m oai_analysis_2_amd.build` "
                       "(there is no CPU fallback in this package)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise OaiError(f"liboai_hip.so does not export {name}")
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().oai_last_error()
        raise OaiError(f"{what or 'liboai_hip'} failed ({status}): {msg.decode() if msg else '?'}")


def int3(v):
    return (C.c_int * 3)(int(v[0]), int(v[1]), int(v[2]))
