"""ctypes binding of libsgx_hip.so (include/sgx_hip.h).

The product has exactly one compute library: the hipcc-built gfx950 shared object in
`super_gradients_amd/csrc/libsgx_hip.so`.  There is no CPU fallback: if the library is missing, or a
tensor that is not on a HIP device reaches a kernel wrapper, we raise.

Two HIP runtimes in one process would make torch's streams/pointers foreign to our kernels.  The wheel's
`torch/lib/libamdhip64.so` and ROCm's `/opt/rocm/lib/libamdhip64.so.7` share the SONAME `libamdhip64.so.7`,
so loading our library AFTER `import torch` makes the dynamic loader bind it to the runtime torch already
loaded; `_check_single_runtime()` verifies that.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_void_p

import torch  # noqa: F401  (must be imported before the library is loaded - see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsgx_hip.so")

_LIB = None
# Set ONLY by tests/emu (kernel-logic tests on host memory against tests/emu/_build/libsgx_emu.so).
# Nothing in the package, bench.py or __graft_entry__ ever sets it.
_TEST_HOST_MODE = False


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("N", "H", "W", "C", "K", "R", "S", "stride", "pad", "Ho", "Wo")] + [
        (n, c_int64) for n in ("x_ld_pix", "x_ld_img", "y_ld_pix", "y_ld_img")
    ]


class LossDesc(ctypes.Structure):
    _fields_ = [
        ("B", c_int32), ("L", c_int32), ("C", c_int32), ("reg_max", c_int32), ("nmax", c_int32),
        ("use_static_assigner", c_int32), ("use_varifocal", c_int32), ("num_levels", c_int32),
        ("level_count", c_int32 * 8), ("w_cls", c_float), ("w_iou", c_float), ("w_dfl", c_float), ("sequential_assignment", c_int32),
    ]


class WtransJob(ctypes.Structure):  # == sgx_wtrans_job
    _fields_ = [("w", ctypes.c_void_p), ("wt", ctypes.c_void_p), ("K", c_int32), ("C", c_int32), ("RS", c_int32), ("T", c_int32), ("taps", ctypes.c_uint8 * 64)]


class WgradJob(ctypes.Structure):  # == sgx_wgrad_job
    _fields_ = [("d", ConvDesc), ("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p)]


class BnReduceReq(ctypes.Structure):  # == sgx_bn_reduce_req
    _fields_ = [("c_lo", c_int32), ("c_hi", c_int32), ("t", ctypes.c_void_p), ("t_ld_pix", ctypes.c_int64), ("t_ld_img", ctypes.c_int64),
                ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("act", c_int32), ("rows", c_int32),
                ("partials", ctypes.c_void_p)]


class QarepPrepJob(ctypes.Structure):  # == sgx_qarep_prep_job
    _fields_ = [("w1", ctypes.c_void_p), ("w1p", ctypes.c_void_p), ("w1pt", ctypes.c_void_p), ("alpha", ctypes.c_void_p), ("K", c_int32), ("C", c_int32),
                ("identity", c_int32), ("pad_", c_int32)]


class FplanesJob(ctypes.Structure):  # == sgx_fplanes_job
    _fields_ = [("src", ctypes.c_void_p), ("planes", ctypes.c_void_p), ("rows", c_int32), ("taps", c_int32), ("ch", c_int32), ("pad_", c_int32)]


class ImageJob(ctypes.Structure):  # == sgx_image_job
    _fields_ = [("src", ctypes.c_void_p), ("h0", c_int32), ("w0", c_int32), ("h", c_int32), ("w", c_int32), ("top", c_int32), ("left", c_int32)]


class NmsDesc(ctypes.Structure):
    _fields_ = [
        ("B", c_int32), ("L", c_int32), ("C", c_int32), ("multi_label", c_int32), ("class_mode", c_int32),
        ("nms_top_k", c_int32), ("max_predictions", c_int32), ("score_threshold", c_float), ("iou_threshold", c_float),
    ]


class MatchDesc(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("B", "P", "nthr", "top_k", "H", "W", "denormalize", "nmax", "cmax")]


_P = c_void_p
_i32, _i64, _f = c_int32, c_int64, c_float
_CD, _LD, _ND = POINTER(ConvDesc), POINTER(LossDesc), POINTER(NmsDesc)

# name -> (restype, argtypes); mirrors include/sgx_hip.h one to one
PROTOTYPES = {
    "sgx_version": (_i32, []),
    "sgx_last_error": (c_char_p, []),
    "sgx_prof_enable": (_i32, [_i32]),
    "sgx_prof_summary": (_i32, [_i32, POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(c_int64)]),
    "sgx_debug_set_tiles": (_i32, [_i32] * 5),
    "sgx_debug_set_variant": (_i32, [_i32]),
    "sgx_conv_tuning_load": (_i32, [POINTER(c_int32), _i32]),
    "sgx_conv_tuning_size": (_i32, []),
    "sgx_conv_set_math": (_i32, [_i32]),
    "sgx_conv_get_math": (_i32, []),
    "sgx_prof_bytes": (_i32, [_i32, POINTER(ctypes.c_double)]),
    "sgx_prof_bound_ms": (_i32, [_i32, ctypes.c_double, ctypes.c_double, POINTER(ctypes.c_double)]),
    "sgx_conv2d_fwd": (_i32, [_CD, _P, _P, _P, _P, _P, _i32, _P, _P]),
    "sgx_conv2d_fwd_stat_blocks": (_i32, [_CD]),
    "sgx_conv2d_bwd_data_workspace": (_i64, [_CD]),
    "sgx_conv2d_bwd_data": (_i32, [_CD, _P, _P, _P, _P, _i32, _P, _i64, _P]),
    "sgx_conv2d_transpose_weights": (_i32, [_CD, _P, _P, _i64, _P]),
    "sgx_conv2d_bwd_data_wt": (_i32, [_CD, _P, _P, _P, _P, _i32, _P]),
    "sgx_conv2d_bwd_data_stat_blocks": (_i32, [_CD, _i32]),
    "sgx_conv2d_bwd_data_wt_req": (_i32, [_CD, _P, _P, _P, _P, _i32, POINTER(BnReduceReq), _i32, _P]),
    "sgx_conv2d_bwd_data_dual_req": (_i32, [_CD, _P, _P, _P, _i64, _i64, _P, _P, _P, _i64, _i64, _f, _P, _P, _i32, POINTER(BnReduceReq), _i32, _P]),
    "sgx_conv2d_transpose_jobs": (_i32, [_CD, _P, _P, _i64, POINTER(WtransJob), _i32, POINTER(c_int32)]),
    "sgx_wtrans_batch": (_i32, [_P, _i32, _P]),
    "sgx_conv2d_fwd_dual_stat_blocks": (_i32, [_CD]),
    "sgx_conv2d_fwd_dual": (_i32, [_CD, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sgx_conv2d_bwd_data_dual": (_i32, [_CD, _P, _P, _P, _i64, _i64, _P, _P, _P, _i64, _i64, _f, _P, _P, _i32, _P]),
    "sgx_qarep_prep_batch": (_i32, [_P, _i32, _P]),
    "sgx_filter_planes_bytes": (_i64, [_i32, _i32, _i32]),
    "sgx_filter_planes_batch": (_i32, [_P, _P, _i32, _P]),
    "sgx_filter_planes_invalidate": (_i32, [_P, _i32]),
    "sgx_filter_planes_scope": (_i32, [_i32]),
    "sgx_debug_set_filter_planes": (_i32, [_i32]),
    "sgx_debug_filter_planes_hits": (_i64, []),
    "sgx_qarep_workspace": (_i64, [_i32, _i32]),
    "sgx_qarep_fwd_finalize": (_i32, [_P, _i32, _i64, _i32, _P, _P, _P, _f, _f, _P, _P, _P, _P, _f, _f, _P, _P, _P, _P, _P, _i64, _P]),
    "sgx_qarep_bwd_reduce": (_i32, [_P, _i64, _P, _i64, _P, _i64, _P, _P, _i64, _i32, _i32, _P, _P]),
    "sgx_qarep_bwd_finalize": (_i32, [_P, _i32, _i64, _i32, _P, _P, _P, _P, _P, _P, _P, _P, _i64, _P]),
    "sgx_qarep_bwd_apply": (_i32, [_P, _i64, _P, _i64, _P, _i64, _P, _P, _P, _P, _i64, _P, _i64, _i64, _i32, _i32, _P]),
    "sgx_hconv2d_fwd": (_i32, [_CD, _P, _P, _P, _P, _i32, _i32, _P, _i64, _i64, _f, _P, _P]),
    "sgx_hconvT2x2_fwd": (_i32, [_i32] * 5 + [_P, _i64, _i64, _P, _P, _P, _i64, _i64, _P]),
    "sgx_hmaxpool_fwd": (_i32, [_i32] * 7 + [_P, _i64, _i64, _P, _i64, _i64, _P]),
    "sgx_hcopy": (_i32, [_P, _i64, _i64, _i32, _P, _i64, _P]),
    "sgx_himage_colsum": (_i32, [_i32, _i32, _i32, _P, _i64, _i64, _f, _P, _P]),
    "sgx_hchannel_gate": (_i32, [_i32, _i32, _i32, _P, _i64, _i64, _P, _i32, _P, _i64, _i64, _P]),
    "sgx_hupsample2x_fwd": (_i32, [_i32, _i32, _i32, _i32, _P, _i64, _i64, _P, _i64, _i64, _P]),
    "sgx_cast_f32_bf16": (_i32, [_P, _i64, _i64, _i32, _P, _i64, _i32, _P]),
    "sgx_hconv_debug_set_tile": (_i32, [_i32] * 3),
    "sgx_conv2d_bwd_weight_workspace": (_i64, [_CD]),
    "sgx_conv2d_bwd_weight": (_i32, [_CD, _P, _P, _P, _P, _P, _i64, _P]),
    "sgx_conv2d_bwd_weight_group_sizes": (_i32, [POINTER(WgradJob), _i32, POINTER(c_int64), POINTER(c_int64)]),
    "sgx_conv2d_bwd_weight_group": (_i32, [POINTER(WgradJob), _i32, _P, _i64, _P, _i64, _P]),
    "sgx_debug_set_wgrad_group": (_i32, [_i32] * 3),
    "sgx_debug_set_wgrad_loop": (_i32, [_i32] * 2),
    "sgx_conv_set_wgrad_math": (_i32, [_i32]),
    "sgx_conv_get_wgrad_math": (_i32, []),
    "sgx_debug_set_wgrad_patch": (_i32, [_i32] * 3),
    "sgx_debug_set_bf3_min_depth": (_i32, [_i32]),
    "sgx_debug_set_pconv_pipe": (_i32, [_i32]),
    "sgx_conv_set_wgrad_lds_reserve": (_i32, [_i32]),
    "sgx_conv_get_wgrad_lds_reserve": (_i32, []),
    "sgx_stream_create_partial": (_i32, [_i32, ctypes.POINTER(ctypes.c_void_p)]),
    "sgx_stream_destroy": (_i32, [ctypes.c_void_p]),
    "sgx_debug_set_nms_split": (_i32, [_i32]),
    "sgx_debug_set_nms_selection": (_i32, [_i32]),
    "sgx_debug_set_igemm_lds_pad": (_i32, [_i32]),
    "sgx_debug_nms_fallback_slot": (_i32, [_P, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)]),
    "sgx_convT2x2_workspace": (_i64, [_i32] * 5),
    "sgx_convT2x2_fwd": (_i32, [_i32] * 5 + [_P, _i64, _i64, _P, _P, _P, _i64, _i64, _P, _i64, _P]),
    "sgx_convT2x2_fwd_wt": (_i32, [_i32] * 5 + [_P, _i64, _i64, _P, _P, _P, _i64, _i64, _P]),
    "sgx_convT2x2_bwd_data": (_i32, [_i32] * 5 + [_P, _i64, _i64, _P, _P, _i64, _i64, _P]),
    "sgx_convT2x2_bwd_weight": (_i32, [_i32] * 5 + [_P, _i64, _i64, _P, _i64, _i64, _P, _P, _P, _i64, _P]),
    "sgx_nchw_to_nhwc": (_i32, [_i32] * 5 + [_P, _P, _P]),
    "sgx_nhwc_to_nchw": (_i32, [_i32] * 4 + [_P, _i64, _i64, _P, _P]),
    "sgx_standardize_u8_hwc": (_i32, [_i32] * 5 + [_P, _f, _P, _P, _P, _P]),
    "sgx_pad_standardize_u8_hwc": (_i32, [_i32, _i32, _i32, _P, _i32, _i32, _i32, _i32, _i32, _f, _P, _P, _P, _P, _P]),
    "sgx_preprocess_u8_hwc": (_i32, [_P, _i32, _i32, _i32, _i32, _i32, _i32, _i32, ctypes.c_double, _P, _P, _P, _P, _P]),
    "sgx_stats_blocks": (_i32, [_i64]),
    "sgx_channel_stats_partial": (_i32, [_P, _i64, _i32, _i64, _P, _P]),
    "sgx_reduce_workspace": (_i64, [_i32, _i32]),
    "sgx_colsum_workspace": (_i64, [_i64, _i32]),
    "sgx_bn_finalize": (_i32, [_P, _i32, _i64, _i32, _P, _P, _f, _f, _P, _P, _P, _P, _P, _P, _P, _i64, _P]),
    "sgx_bn_reduce_sums": (_i32, [_P, _i32, _i32, _P, _P, _i64, _P]),
    "sgx_bn_finalize_sums": (_i32, [_P, _i64, _i32, _P, _P, _f, _f, _P, _P, _P, _P, _P, _P, _P]),
    "sgx_bn_bwd_finalize_sums": (_i32, [_P, _P, _i64, _i32, _P, _P, _P, _P, _P, _P, _P]),
    "sgx_bn_eval_scale_shift": (_i32, [_i32, _P, _P, _P, _P, _f, _P, _P, _P]),
    "sgx_affine_act_fwd": (_i32, [_P, _i64, _P, _P, _P, _i64, _f, _P, _P, _i64, _f, _P, _i64, _i64, _i32, _i32, _P, _P]),
    "sgx_bn_bwd_reduce": (_i32, [_P, _i64, _P, _i64, _P, _P, _P, _i64, _i32, _i32, _P, _P]),
    "sgx_bn_bwd_finalize": (_i32, [_P, _i32, _i64, _i32, _P, _P, _P, _P, _P, _P, _P, _i64, _P]),
    "sgx_bn_bwd_apply": (_i32, [_P, _i64, _P, _i64, _P, _P, _P, _P, _i64, _P, _i64, _i64, _i32, _i32, _P]),
    "sgx_bn_set_fused_finalize": (_i32, [_i32]),
    "sgx_bn_get_fused_finalize": (_i32, []),
    "sgx_dot_workspace": (_i64, [_i64, _i32]),
    "sgx_dot": (_i32, [_P, _i64, _P, _i64, _i64, _i32, _f, _P, _i32, _P, _i64, _P]),
    "sgx_axpy": (_i32, [_P, _i64, _f, _P, _P, _i64, _i64, _i32, _i32, _P]),
    "sgx_relu_bwd": (_i32, [_P, _i64, _P, _i64, _P, _i64, _i64, _i32, _P]),
    "sgx_relu_bwd_bn_reduce": (_i32, [_P, _i64, _P, _i64, _P, _i64, _P, _P, _i64, _i64, _i32, _P, _P]),
    "sgx_colsum": (_i32, [_P, _i64, _i64, _i32, _i64, _i64, _P, _i32, _P, _P]),
    "sgx_maxpool_fwd": (_i32, [_i32] * 7 + [_P, _i64, _i64, _P, _i64, _i64, _P, _P]),
    "sgx_maxpool_bwd": (_i32, [_i32] * 7 + [_P, _P, _i64, _i64, _P, _i64, _i64, _i32, _P]),
    "sgx_avgpool_fwd": (_i32, [_i32] * 3 + [_P, _i64, _i64, _P, _P]),
    "sgx_avgpool_bwd": (_i32, [_i32] * 3 + [_P, _P, _i64, _i64, _P]),
    "sgx_dual_affine_act_fwd": (_i32, [_P, _i64, _P, _P, _P, _i64, _P, _P, _P, _i64, _f, _P, _P, _i64, _i64, _i32, _i32, _P]),
    "sgx_dual_affine_act_bwd": (_i32, [_P, _i64, _P, _i64, _P, _P, _P, _i64, _P, _P, _P, _i64, _i64, _i32, _i32, _P]),
    "sgx_dual_affine_act_bwd_reduce": (_i32, [_P, _i64, _P, _i64, _P, _P, _P, _P, _i64, _P, _P, _P, _P, _i64, _i64, _i32, _i32, _P, _P]),
    "sgx_image_colsum_workspace": (_i64, [_i32] * 3),
    "sgx_image_colsum": (_i32, [_i32] * 3 + [_P, _i64, _i64, _P, _i64, _i64, _f, _P, _i32, _P, _P, _i64, _P]),
    "sgx_channel_gate": (_i32, [_i32] * 3 + [_P, _i64, _i64, _P, _i32, _P, _f, _P, _i64, _i64, _i32, _P]),
    "sgx_upsample2x_fwd": (_i32, [_i32] * 4 + [_P, _i64, _i64, _P, _i64, _i64, _P]),
    "sgx_upsample2x_bwd": (_i32, [_i32] * 4 + [_P, _i64, _i64, _P, _i64, _i64, _i32, _P]),
    "sgx_dfl_decode": (_i32, [_i32] * 4 + [_P] * 6 + [_P]),
    "sgx_targets_index": (_i32, [_P, _i32, _i32, _i32, _P, _P, _P, _P]),
    "sgx_ppyoloe_loss_workspace": (_i64, [_LD]),
    "sgx_ppyoloe_loss_fwd": (_i32, [_LD] + [_P] * 14 + [_P, _i64, _P]),
    "sgx_ppyoloe_loss_finalize": (_i32, [_P, _f, _f, _f, _f, _P, _P, _P]),
    "sgx_scale_by_device_scalar": (_i32, [_P, _P, _P, _P, _i64, _P]),
    "sgx_nms_workspace": (_i64, [_ND]),
    "sgx_nms": (_i32, [_ND, _P, _P, _P, _P, _P, _P, _P, _i64, _P]),
    "sgx_detection_match": (_i32, [POINTER(MatchDesc)] + [_P] * 12),
    "sgx_detection_unmap": (_i32, [_P, _P, _i32, _i32, _P, _i32, _P, _P]),
    "sgx_softmax_ce_fwd_bwd": (_i32, [_i32, _i32, _P, _P, _f, _P, _i32, _i32, _P, _P, _P]),
    "sgx_adamw_step": (_i32, [_P, _P, _P, _P, _i64, _f, _f, _f, _f, _i32, _P, _P, _i32, _P, _P]),
    "sgx_sgd_step": (_i32, [_P, _P, _P, _i64, _f, _f, _f, _i32, _i32, _P, _P, _i32, _P]),
    "sgx_ema_update": (_i32, [_P, _P, _i64, _f, _P]),
    "sgx_fill": (_i32, [_P, _i64, _f, _P]),
}


def bind(cdll):
    """Attach restype/argtypes for every symbol include/sgx_hip.h declares; raises if one is missing."""
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(cdll, name)  # AttributeError -> the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    return cdll


def _check_single_runtime():
    try:
        with open("/proc/self/maps") as f:
            libs = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        return
    if len(libs) > 1:
        raise RuntimeError(
            "two HIP runtimes are mapped in this process (%s): import torch before loading libsgx_hip.so" % sorted(libs)
        )


def lib():
    """The bound library.  Fails loudly when it has not been built (python __graft_entry__.py / csrc/build.py)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). Build it with "
                "`python super_gradients_amd/csrc/build.py` (hipcc --offload-arch=gfx950)."
            )
        _LIB = bind(ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL))
        _check_single_runtime()
        mode = os.environ.get("SGX_CONV_MATH")   # "fp32" | "bf16x3" | "auto" (kernels.set_conv_math); unset = the library default
        if mode:
            modes = {"fp32": 0, "bf16x3": 1, "auto": 2, "patch": 3, "patch_auto": 4, "patch_bf3": 5}
            if mode not in modes:
                raise RuntimeError(f"SGX_CONV_MATH={mode!r}: expected one of {sorted(modes)}")
            _LIB.sgx_conv_set_math(modes[mode])
        var = os.environ.get("SGX_CONV_VARIANT")  # measurement switch of the conv kernels (sgx_debug_set_variant; 7 = the 16-deep loop)
        if var:
            _LIB.sgx_debug_set_variant(int(var))
        wgg = os.environ.get("SGX_WGRAD_GROUP")  # measurement switch of the grouped weight gradient: "rounds,item_mflop,xcd_order" (0 = default)
        if wgg:
            _LIB.sgx_debug_set_wgrad_group(*[int(v) for v in wgg.split(",")])
        # measurement switches of the weight-gradient loop: 32-pixel slabs (small tiles); ablation bits (honoured by -DSGX_WGRAD_LAB builds only);
        # one tile shape for every layer "bnk,bj"
        if os.environ.get("SGX_WGRAD_SLAB") == "32" or os.environ.get("SGX_WGRAD_ABLATE") or os.environ.get("SGX_WGRAD_PF"):
            _LIB.sgx_debug_set_wgrad_loop(int(os.environ.get("SGX_WGRAD_SLAB") == "32") + 2 * int(os.environ.get("SGX_WGRAD_PF") == "1"),
                                          int(os.environ.get("SGX_WGRAD_ABLATE", "0")))
        wgm = os.environ.get("SGX_WGRAD_MATH")  # "fp32" | "bf16x3" (slab loop only) | "patch" (the default: bf16x3 + the patch kernel)
        if wgm:
            if wgm not in WGRAD_MATH:
                raise RuntimeError(f"SGX_WGRAD_MATH={wgm!r}: expected one of {sorted(WGRAD_MATH)}")
            _LIB.sgx_conv_set_wgrad_math(WGRAD_MATH[wgm])
        if os.environ.get("SGX_BF3_MIN_DEPTH"):  # measurement: depth (taps x channels) from which a problem runs in bf16x3 arithmetic
            _LIB.sgx_debug_set_bf3_min_depth(int(os.environ["SGX_BF3_MIN_DEPTH"]))
        if os.environ.get("SGX_FILTER_PLANES"):  # measurement: 0 = every bf16x3 launch splits its filter while staging (round 5)
            _LIB.sgx_debug_set_filter_planes(int(os.environ["SGX_FILTER_PLANES"]))
        if os.environ.get("SGX_PCONV_PIPE"):  # measurement: second fragment set for the patch kernel's 32-filter tiles
            _LIB.sgx_debug_set_pconv_pipe(int(os.environ["SGX_PCONV_PIPE"]))
        if os.environ.get("SGX_WGRAD_LDS_RESERVE"):  # KB of every CU's LDS the weight-gradient kernels leave to the main stream
            _LIB.sgx_conv_set_wgrad_lds_reserve(int(os.environ["SGX_WGRAD_LDS_RESERVE"]))
        if os.environ.get("SGX_WGRAD_PATCH"):  # measurement: "item_mflop,kb,min_fill_pct"
            _LIB.sgx_debug_set_wgrad_patch(*[int(v) for v in os.environ["SGX_WGRAD_PATCH"].split(",")])
        if os.environ.get("SGX_WGRAD_TILE"):
            _LIB.sgx_debug_set_tiles(0, 0, *[int(v) for v in os.environ["SGX_WGRAD_TILE"].split(",")], 0)
        if os.environ.get("SGX_FUSED_FINALIZE") == "0":  # measurement switch: two-launch BatchNorm / column-sum finalize (default: one launch)
            _LIB.sgx_bn_set_fused_finalize(0)
        # per-problem (tile, variant) table measured by tools/conv_tune.py --emit-table: SGX_CONV_TUNING=<json> ("" / "0" = none),
        # default csrc/conv_tuning_gfx950.json when it has been committed
        tune = os.environ.get("SGX_CONV_TUNING")
        if tune is None and os.path.exists(DEFAULT_TUNING):
            tune = DEFAULT_TUNING
        if tune and tune != "0":
            load_conv_tuning(tune, _LIB)
    return _LIB


WGRAD_MATH = {"fp32": 0, "bf16x3": 1, "patch": 2}
DEFAULT_TUNING = os.path.join(_HERE, "csrc", "conv_tuning_gfx950.json")
TUNE_FIELDS = ("kind", "N", "H", "W", "C", "K", "R", "stride", "pad", "bm", "bn", "variant")


def load_conv_tuning(path_or_entries, library=None) -> int:
    """Install a per-problem tuning table (sgx_conv_tuning_load).  `path_or_entries`: a JSON file {"entries": [{kind: "fwd"|"dgrad", N, H, W,
    C, K, R, stride, pad, bm, bn, variant}, ...]} as tools/conv_tune.py --emit-table writes it, or such a list; [] clears.  -> entries installed"""
    import json

    entries = path_or_entries
    if isinstance(entries, str):
        with open(entries) as f:
            entries = json.load(f)["entries"]
    flat = []
    for e in entries:
        flat += [{"fwd": 0, "dgrad": 1, "wgrad": 2}[e["kind"]]] + [int(e[k]) for k in TUNE_FIELDS[1:]]  # wgrad: bm / bn / variant = filter tile / column tile / split target
    arr = (c_int32 * max(len(flat), 1))(*flat)
    L = library if library is not None else lib()
    rc = L.sgx_conv_tuning_load(arr, len(entries))
    if rc != 0:
        msg = L.sgx_last_error()
        raise RuntimeError(f"sgx_conv_tuning_load failed with status {rc}: {msg.decode() if msg else ''}")
    return len(entries)


class SgxError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = lib().sgx_last_error()
        raise SgxError(f"{what} failed with status {rc}: {msg.decode() if msg else ''}")


_PTR_DTYPES = frozenset((torch.float32, torch.float64, torch.int32, torch.int64, torch.uint8, torch.bfloat16))  # bf16: the half-precision inference path


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must live on the HIP device."""
    if t is None:
        return None
    if not (t.is_cuda or _TEST_HOST_MODE):
        raise SgxError("libsgx_hip kernels need tensors on the HIP device (got a CPU tensor); there is no CPU fallback")
    if t.dtype not in _PTR_DTYPES:
        raise SgxError(f"unsupported dtype {t.dtype}")
    return t.data_ptr()


def _stream_slow():
    return torch.cuda.current_stream().cuda_stream


def _stream_fast():
    return torch._C._cuda_getCurrentRawStream(-1)  # -1: the current device


def _stream_probe():
    """First launch: adopt the raw-handle call only if this torch build has it and it agrees with the public API."""
    global _stream_impl
    slow = _stream_slow()
    try:
        ok = _stream_fast() == slow
    except Exception:  # noqa: BLE001
        ok = False
    _stream_impl = _stream_fast if ok else _stream_slow
    return slow


_stream_impl = _stream_probe


def stream():
    """The current HIP stream of the current device as a raw handle (what torch.cuda.current_stream().cuda_stream returns, without
    building the Python Stream object: this runs once per kernel launch, ~1100 times per train step)."""
    if _TEST_HOST_MODE:
        return None
    return _stream_impl()
