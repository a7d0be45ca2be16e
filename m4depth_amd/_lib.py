"""ctypes binding to libm4depth_hip.so (the C ABI of include/m4depth_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call
returns a HIP error, this raises.  PyTorch is used only as the owner of device
memory and streams (``tensor.data_ptr()``, ``torch.cuda.current_stream()``).
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# M4D_LIB_FLAVOUR=experiments selects the `make EXPERIMENTS=1` library (libm4depth_hip_exp.so: the product kernels + the
# measured-and-not-dispatched ones of tools/experiments/csrc with their process-global selectors); anything else = the product
LIB_FLAVOUR = os.environ.get("M4D_LIB_FLAVOUR", "product")
LIB_PATH = os.path.join(_HERE, "libm4depth_hip_exp.so" if LIB_FLAVOUR == "experiments" else "libm4depth_hip.so")

ABI_VERSION = 6               # M4D_ABI_VERSION of include/m4depth_hip.h this binding was written for

_c_fp = ctypes.c_void_p       # device pointers travel as void*
_c_int = ctypes.c_int
_c_f = ctypes.c_float

class NormLevel(ctypes.Structure):
    """m4d_norm_level of include/m4depth_hip.h (one map of m4d_normalize_levels)."""
    _fields_ = [("x", ctypes.c_void_p), ("out", ctypes.c_void_p), ("pixels", ctypes.c_longlong), ("C", ctypes.c_int),
                ("nbre_cuts", ctypes.c_int)]


class ResetLevel(ctypes.Structure):
    """m4d_reset_level of include/m4depth_hip.h (one level of m4d_pyramid_reset)."""
    _fields_ = [("features", ctypes.c_void_p), ("state_features", ctypes.c_void_p), ("depth_state", ctypes.c_void_p),
                ("parallax", ctypes.c_void_p), ("depth", ctypes.c_void_p), ("other", ctypes.c_void_p),
                ("h", ctypes.c_int), ("w", ctypes.c_int), ("C", ctypes.c_int), ("nbre_cuts", ctypes.c_int),
                ("parallax_value", ctypes.c_float)]


# name -> argtypes; mirrors include/m4depth_hip.h one for one.
_SIGNATURES = {
    "m4d_backproject_fwd": [_c_fp, _c_fp, ctypes.POINTER(_c_int), _c_fp, _c_fp],
    "m4d_backproject_bwd": [_c_fp, _c_fp, _c_fp, ctypes.POINTER(_c_int), _c_fp, _c_fp, _c_fp],
    "m4d_dense_image_warp": [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_fp, _c_fp],
    "m4d_interpolate_bilinear": [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_fp, _c_fp],
    "m4d_parallax2depth": [_c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_depth2parallax": [_c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_prev_d2para": [_c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_recompute_depth": [_c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_reproject_flow": [_c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_fp, _c_fp, _c_fp, _c_fp],
    "m4d_dscv_fwd": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp,
                     _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                     _c_fp, _c_int, _c_fp, _c_fp, _c_int, _c_f, _c_fp, _c_fp],
    "m4d_dscv_sncv_fwd": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp,
                          _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                          _c_fp, _c_int, _c_fp, _c_fp, _c_int, _c_f, _c_int, _c_fp, _c_int, _c_fp],
    "m4d_sncv_fwd": [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_int, _c_fp],
    "m4d_dscv_bwd": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp,
                     _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                     _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "m4d_sncv_bwd": [_c_fp, _c_fp, _c_fp, _c_int, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                     _c_int, _c_f, _c_fp, _c_fp, _c_fp],
    "m4d_resize_bilinear_v1_bwd": [_c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_fp],
    "m4d_level_post_bwd": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_f,
                           _c_fp, _c_fp],
    "m4d_normalize_cuts_bwd": [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_bias_act_bwd": [_c_fp, _c_fp, ctypes.c_longlong, _c_int, _c_f, _c_fp, _c_fp, _c_fp, _c_fp],
    "m4d_pack_conv_weights": [_c_fp, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_pack_conv_weights_lat": [_c_fp, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_loss_level_fwd": [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_fp, _c_fp],
    "m4d_loss_level_bwd": [_c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_decode_rgb8_resize": [_c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_decode_depth_resize": [_c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, ctypes.POINTER(_c_int),
                                _c_fp, _c_fp],
    "m4d_normalize_cuts": [_c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_resize_bilinear_v1": [_c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_fp],
    "m4d_resize_nearest": [_c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_bias_act": [_c_fp, _c_fp, ctypes.c_longlong, _c_int, _c_f, _c_fp, _c_fp],
    "m4d_dinl_fwd": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_fp, _c_fp],
    "m4d_dinl_fwd_padded": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_fp,
                            _c_int, _c_int, _c_int, _c_int, _c_fp],
    "m4d_bias_act_padded": [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_fp],
    "m4d_conv3x3_bias_act": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_fp],
    "m4d_conv3x3_bias_act_ws": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp,
                                _c_fp, ctypes.c_longlong, _c_fp],
    "m4d_conv3x3s_bias_act_ws": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f,
                                 _c_fp, _c_fp, ctypes.c_longlong, _c_fp],
    "m4d_conv3x3_small_bias_act": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_fp],
    "m4d_conv3x3s_small_bias_act": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_fp],
    "m4d_conv3x3_small6_bias_act": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_fp],
    "m4d_conv3x3_lat": [_c_fp, _c_int, ctypes.c_longlong, _c_fp, _c_f, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f,
                        _c_int, _c_int, _c_int, _c_fp, ctypes.c_longlong, _c_fp],
    "m4d_conv3x3s_lat": [_c_fp, _c_int, ctypes.c_longlong, _c_fp, _c_f, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int,
                         _c_f, _c_int, _c_int, _c_int, _c_fp, ctypes.c_longlong, _c_fp],
    "m4d_partial_finish": [_c_fp, _c_int, ctypes.c_longlong, _c_fp, _c_f, ctypes.c_longlong, _c_int, _c_fp, _c_fp],
    "m4d_conv3x3_wino_bias_act": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_fp],
    "m4d_conv3x3_wino2_bias_act": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_fp],
    "m4d_conv3x3_wino6_bias_act": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_fp],
    "m4d_conv3x3_wino6_bias_act_k": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_int, _c_fp],
    "m4d_conv3x3_wino6_bias_act_ks": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_int,
                                      _c_int, _c_int, _c_fp],
    "m4d_enc_head_fwd": [_c_fp, _c_int, ctypes.c_longlong, ctypes.c_longlong, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int,
                         _c_fp, _c_fp, _c_fp],
    "m4d_enc_level0_stats": [_c_fp, _c_int, ctypes.c_longlong, ctypes.c_longlong, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_fp, _c_fp, _c_fp,
                             _c_fp],
    "m4d_enc_level0_apply": [_c_fp, _c_int, ctypes.c_longlong, ctypes.c_longlong, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_f, _c_fp,
                             _c_fp, _c_f, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_enc_level0_fwd": [_c_fp, _c_int, ctypes.c_longlong, ctypes.c_longlong, _c_fp, _c_fp, _c_fp, _c_fp, _c_f, _c_fp, _c_fp, _c_f,
                           _c_int, _c_int, _c_int, _c_fp, _c_fp, _c_fp],
    "m4d_conv3x3s2_dinl_bias_act": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_f, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int,
                                    _c_f, _c_fp, _c_fp],
    "m4d_refiner_tail": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_f,
                         _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "m4d_refiner_tail6": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_f,
                          _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
    "m4d_depth_metrics": [_c_fp, _c_fp, ctypes.c_longlong, _c_f, _c_fp, _c_fp, _c_fp, _c_f, _c_fp, _c_fp],
    "m4d_depth_metrics_strided": [_c_fp, ctypes.c_longlong, ctypes.c_longlong, _c_fp, ctypes.c_longlong, _c_f, _c_fp, _c_fp, _c_fp,
                                  _c_f, _c_fp, _c_fp],
    "m4d_level_pre": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int,
                      _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_f, _c_fp, _c_fp],
    "m4d_level_pre_normalize": [_c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int,
                                _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_f, _c_fp,
                                _c_fp, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_level_front_supported": [_c_int, _c_int, _c_int, _c_int, _c_int],
    "m4d_pyramid_reset_supported": [_c_int, _c_int],
    "m4d_pyramid_reset": [ctypes.POINTER(ResetLevel), _c_int, _c_int, _c_fp],
    "m4d_normalize_levels": [ctypes.POINTER(NormLevel), _c_int, _c_fp],
    "m4d_level_front_small_supported": [_c_int, _c_int, _c_int, _c_int],
    "m4d_level_front_small": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_fp, _c_int, _c_fp, _c_fp, _c_fp,
                              _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_int, _c_f, _c_fp],
    "m4d_level_front": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_fp, _c_int, _c_fp, _c_fp, _c_fp,
                        _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_int, _c_f, _c_fp],
    "m4d_level_front_r": [_c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_fp, _c_int, _c_fp, _c_fp, _c_fp,
                          _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_int, _c_f, _c_fp],
    "m4d_conv3x3_wgrad": [_c_fp, _c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, ctypes.c_longlong, _c_fp, _c_fp],
    "m4d_dilate2": [_c_fp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_fp, _c_fp],
    "m4d_camera_pyramid": [_c_fp, _c_fp, _c_int, _c_int, _c_fp, _c_fp, _c_fp],
    "m4d_level_post": [_c_fp, _c_fp, _c_int, _c_fp, _c_fp, _c_fp, _c_int, _c_int, _c_int, _c_f,
                       _c_fp, _c_fp, _c_fp, _c_fp, _c_fp],
}

# include/m4depth_hip_experiments.h: present only in a `make EXPERIMENTS=1` build of the library (the launch tape and the
# selectors of the not-dispatched Winograd kernels); bound when the library has them, ``has_experiments`` says whether
_EXPERIMENT_SIGNATURES = {"m4d_tape_begin": [], "m4d_tape_end": [], "m4d_tape_length": [_c_int], "m4d_tape_replay": [_c_int, _c_fp],
                          "m4d_tape_free": [_c_int]}
_EXPERIMENT_VOID_SIGNATURES = {"m4d_wino6_set_variant": [_c_int], "m4d_wino6_set_half_tile_max_workgroups": [_c_int]}
EXPERIMENT_SYMBOLS = list(_EXPERIMENT_SIGNATURES) + list(_EXPERIMENT_VOID_SIGNATURES)

_LL_SIGNATURES = {"m4d_launch_count": [], "m4d_wino6_persistent_min_units": [], "m4d_conv3x3_workspace_floats": [_c_int, _c_int, _c_int, _c_int],
                  "m4d_dinl_workspace_floats": [_c_int, _c_int], "m4d_metrics_workspace_bytes": [],
                  "m4d_bias_act_bwd_workspace_floats": [ctypes.c_longlong, _c_int], "m4d_loss_workspace_floats": [],
                  "m4d_conv3x3_wgrad_workspace_floats": [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int]}
_VOID_SIGNATURES = {"m4d_dscv_set_variant": [_c_int], "m4d_dscv_set_fallback_counter": [_c_fp],
                    "m4d_dscv_set_ablation": [_c_int], "m4d_dscv_set_stamps": [_c_fp], "m4d_wino_set_stamps": [_c_fp],
                    "m4d_front_set_stamps": [_c_fp], "m4d_wino6_set_stamps": [_c_fp]}

EXPORTED_SYMBOLS = ["m4d_abi_version", "m4d_build_info"] + list(_SIGNATURES) + list(_VOID_SIGNATURES) + list(_LL_SIGNATURES)


def _load():
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"m4depth_amd: native library not found at {LIB_PATH}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C m4depth_amd/csrc`. "
            "There is no CPU / PyTorch fallback for the hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    lib.m4d_abi_version.restype = _c_int
    lib.m4d_abi_version.argtypes = []
    lib.m4d_build_info.restype = ctypes.c_char_p
    lib.m4d_build_info.argtypes = []
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so is stale: loud by design
        fn.restype = _c_int
        fn.argtypes = args
    for name, args in _LL_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_longlong
        fn.argtypes = args
    for name, args in _VOID_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = None
        fn.argtypes = args
    if hasattr(lib, "m4d_tape_begin"):
        for name, args in _EXPERIMENT_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = _c_int
            fn.argtypes = args
        for name, args in _EXPERIMENT_VOID_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = None
            fn.argtypes = args
    if lib.m4d_abi_version() != ABI_VERSION:
        raise ImportError(f"m4depth_amd: ABI mismatch, library reports {lib.m4d_abi_version()}, binding expects {ABI_VERSION}")
    return lib


lib = _load()
has_experiments = hasattr(lib, "m4d_tape_begin")      # a `make EXPERIMENTS=1` build (include/m4depth_hip_experiments.h)


def require_experiments(what):
    if not has_experiments:
        raise RuntimeError(f"{what} needs the experiments build of the library: make -C m4depth_amd/csrc EXPERIMENTS=1 "
                           "(-> libm4depth_hip_exp.so, sources in tools/experiments/csrc) and M4D_LIB_FLAVOUR=experiments in "
                           "the environment; the product library exports only the kernels it dispatches")


def build_info() -> str:
    return lib.m4d_build_info().decode()


def stream_ptr():
    """The current PyTorch HIP stream as a void* for the C ABI."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dptr(t, name="tensor", dtype=torch.float32):
    """Device pointer of a dense CUDA(ROCm) tensor; None -> NULL."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: m4depth_amd ops run on the MI355X only (tensor is on {t.device}); "
                           "there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a dense row-major (NHWC) tensor, got strides {t.stride()}")
    return ctypes.c_void_p(t.data_ptr())


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"m4depth_amd: {what} failed with HIP error {rc}"
                           + (" (hipErrorInvalidValue: bad argument)" if rc == 1 else ""))


# Framework copies inside a hipGraph capture: refused on the inference path (every node of its graph is a library kernel), allowed
# inside ``framework_copies_in_capture()`` -- the TRAINING step's capture, whose autograd graph is made of framework nodes anyway
# (slice copies, gradient accumulation) around the library's kernels.
_capture_copies_ok = [False]


class framework_copies_in_capture:
    def __enter__(self):
        self._old = _capture_copies_ok[0]
        _capture_copies_ok[0] = True
        return self

    def __exit__(self, *exc):
        _capture_copies_ok[0] = self._old
        return False


def as_f32(t, name):
    """Dense float32 device tensor (copies only when a conversion is needed)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if t.dtype == torch.float32 and t.is_contiguous():
        return t
    if t.is_cuda and not _capture_copies_ok[0] and torch.cuda.is_current_stream_capturing():
        # a conversion here would put a framework copy kernel into the captured graph (round 4: ~43 per batch-32 step, the
        # [:, t] slices of batch-major rot / trans): the capturing callers hand over dense frame-major tensors instead
        raise RuntimeError(f"{name}: a non-contiguous / non-float32 tensor (shape {tuple(t.shape)}, strides {t.stride()}, "
                           f"{t.dtype}) reached a kernel wrapper inside a hipGraph capture; it would be copied by a framework "
                           "kernel inside the graph -- make it dense before capturing")
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
