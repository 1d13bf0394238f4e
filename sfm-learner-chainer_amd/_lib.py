"""ctypes binding of libsfmwarp.so (include/sfmwarp.h).  No torch types cross this boundary:
every tensor is handed over as a raw device pointer plus explicit sizes.

The library is REQUIRED: there is no CPU or PyTorch fallback.  If it is missing, importing
this module raises, and so does every operator of the package.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch ships its own libamdhip64; libsfmwarp.so must bind to that SAME runtime instance (device
# pointers, streams and events are shared with torch), so torch has to be loaded first.  Loading
# /opt/rocm's copy first and torch's afterwards leaves this library without a visible device.
import torch  # noqa: F401

SFM_MAX_SCALES = 8
SFM_MAX_SRC = 8
SFM_ABI_VERSION = 5
SFM_LAYOUT_PLANAR, SFM_LAYOUT_HWC = 0, 1
SFM_PROJECTION_FAST, SFM_PROJECTION_REFERENCE_ORDER = 0, 1
PROJECTIONS = {None: SFM_PROJECTION_FAST, "fast": SFM_PROJECTION_FAST, "reference_order": SFM_PROJECTION_REFERENCE_ORDER}

SMOOTH_NONE, SMOOTH_SECOND_ORDER, SMOOTH_EDGE_AWARE = 0, 1, 2
SMOOTH_MODES = {None: SMOOTH_NONE, "none": SMOOTH_NONE, "second_order": SMOOTH_SECOND_ORDER,
                "edge_aware": SMOOTH_EDGE_AWARE}

ERR_NULL, ERR_SHAPE, ERR_CONFIG, ERR_WORKSPACE = -1, -2, -3, -4

_FP = C.c_void_p   # device float*


class SfmLossDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("norm_B", C.c_int32), ("n_src", C.c_int32), ("n_scales", C.c_int32),
        ("H", C.c_int32 * SFM_MAX_SCALES), ("W", C.c_int32 * SFM_MAX_SCALES),
        ("smooth_reg", C.c_float), ("exp_reg", C.c_float), ("ssim_rate", C.c_float),
        ("smooth_mode", C.c_int32),
        ("tgt", _FP * SFM_MAX_SCALES), ("src", _FP * SFM_MAX_SCALES), ("disp", _FP * SFM_MAX_SCALES),
        ("mask_logits", _FP * SFM_MAX_SCALES), ("intrinsics", _FP), ("pose", _FP * SFM_MAX_SRC),
        ("d_disp", _FP * SFM_MAX_SCALES), ("d_pose", _FP * SFM_MAX_SRC), ("d_mask", _FP * SFM_MAX_SCALES),
        ("d_src", _FP * SFM_MAX_SCALES),
        ("image_layout", C.c_int32),
        ("warped", _FP * SFM_MAX_SCALES),
        ("projection", C.c_int32),
    ]


# SFMWARP_LIB selects another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("SFMWARP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsfmwarp.so")

# every symbol declared in include/sfmwarp.h: name -> (restype, argtypes)
_I, _V, _Z = C.c_int, C.c_void_p, C.c_size_t
SYMBOLS = {
    "sfm_abi_version": (_I, []),
    "sfm_last_error": (C.c_char_p, []),
    "sfm_pose_proj_fwd": (_I, [_FP, _FP, _FP, _I, _V]),
    "sfm_pose_proj_bwd": (_I, [_FP, _FP, _FP, _FP, _I, _V]),
    "sfm_warp_fwd": (_I, [_FP, _FP, _I, _FP, _FP, _FP, _I, _I, _I, _I, _V]),
    "sfm_warp_bwd_workspace_bytes": (_Z, [_I, _I, _I]),
    "sfm_warp_bwd": (_I, [_FP, _FP, _I, _FP, _FP, _FP, _FP, _FP, _FP, _V, _Z, _I, _I, _I, _I, _V]),
    "sfm_sampler_fwd": (_I, [_FP, _FP, _FP, _I, _I, _I, _I, _I, _I, _V]),
    "sfm_sampler_bwd": (_I, [_FP, _FP, _FP, _FP, _FP, _I, _I, _I, _I, _I, _I, _V]),
    "sfm_sampler_interp_fwd": (_I, [_FP, _FP, _FP, _I, _I, _I, _I, _I, _I, _V]),
    "sfm_sampler_interp_bwd": (_I, [_FP, _FP, _FP, _FP, _FP, _I, _I, _I, _I, _I, _I, _V]),
    "sfm_loss_workspace_bytes": (_Z, [C.POINTER(SfmLossDesc)]),
    "sfm_loss_fwd": (_I, [C.POINTER(SfmLossDesc), _FP, _V, _Z, _V]),
    "sfm_loss_bwd": (_I, [C.POINTER(SfmLossDesc), C.c_float, _V, _Z, _V]),
    "sfm_loss_fwd_bwd": (_I, [C.POINTER(SfmLossDesc), _FP, _V, _Z, _V]),
    "sfm_step_fwd": (_I, [_FP, _FP, C.POINTER(SfmLossDesc), _FP, _V, _Z, _V]),
    "sfm_step_fwd_bwd": (_I, [_FP, _FP, C.POINTER(SfmLossDesc), _FP, _V, _Z, _V]),
    "sfm_loss_plan_info": (_I, [C.POINTER(SfmLossDesc), _I, _I, C.POINTER(C.c_int), _I]),
    "sfm_loss_profile_events": (_I, [_V, _V]),
    "sfm_loss_debug_trace": (_I, [_V]),
    "sfm_loss_variant": (_I, [_I]),
    "sfm_resize_fwd": (_I, [_FP, _FP, _I, _I, _I, _I, _I, _I, _V]),
    "sfm_disp_act_fwd": (_I, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_longlong), _I, _V]),
    "sfm_disp_act_bwd": (_I, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_longlong), _I, _V]),
    "sfm_augment_fwd": (_I, [_FP, _FP, _FP, _I, _I, _I, _I, _I, _V]),
    "sfm_pyramid_fwd": (_I, [_FP, C.POINTER(C.c_void_p), _I, _I, _I, _I, _I, _V]),
    "sfm_pyramid_variant": (_I, [_I]),
    "sfm_pyramid_hwc_fwd": (_I, [_FP, C.POINTER(C.c_void_p), _I, _I, _I, _I, _I, _V]),
    "sfm_pyramid_pair_hwc_fwd": (_I, [_FP, _FP, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I, _I, _I, _I, _I, _V]),
}


class SfmWarpError(RuntimeError):
    """A launch failed inside libsfmwarp (positive return code = hipError_t)."""


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libsfmwarp.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C sfm-learner-chainer_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.sfm_abi_version()
    if got != SFM_ABI_VERSION:
        raise ImportError("libsfmwarp.so has ABI version %d, this package expects %d" % (got, SFM_ABI_VERSION))
    return lib


lib = _load()


def last_error() -> str:
    msg = lib.sfm_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(code: int) -> None:
    """Maps the C return convention onto the reference's classes of failure: bad shapes /
    dtypes are TypeError (Chainer's type_check.InvalidType is a TypeError-like check failure),
    inconsistent configuration is ValueError, launch failures are RuntimeError."""
    if code == 0:
        return
    msg = last_error()
    if code in (ERR_NULL, ERR_SHAPE):
        raise TypeError(msg)
    if code in (ERR_CONFIG, ERR_WORKSPACE):
        raise ValueError(msg)
    raise SfmWarpError("%s (code %d)" % (msg, code))
