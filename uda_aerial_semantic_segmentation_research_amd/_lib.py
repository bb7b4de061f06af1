"""ctypes binding of libudaseg_hip.so (the C-ABI declared in include/udaseg.h).

There is no CPU fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C <pkg>/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UDASEG_LIB", os.path.join(_HERE, "libudaseg_hip.so"))   # UDASEG_LIB: an alternative build (tuning)

ACT_NONE = 0
ACT_LEAKY = 1  # slope 0 => ReLU


class ConvDesc(C.Structure):
    """udaseg_conv_desc"""
    _fields_ = [(k, C.c_int) for k in ("n", "hi", "wi", "ci", "ho", "wo", "co", "kh", "kw", "stride", "pad")]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_D = C.POINTER(ConvDesc)

# name -> (restype, argtypes); every symbol include/udaseg.h declares
SIGNATURES = {
    "udaseg_version": (_I, []),
    "udaseg_set_option": (_I, [_I, _I]),
    "udaseg_get_option": (_I, [_I]),
    "udaseg_option_count": (_I, []),
    "udaseg_option_name": (C.c_char_p, [_I]),
    "udaseg_option_epoch": (_I, []),
    "udaseg_last_error": (C.c_char_p, []),
    "udaseg_device_count": (_I, []),
    "udaseg_memset_async": (_I, [_P, _I, C.c_size_t, _P]),
    "udaseg_stream_wait": (_I, [_P, _P]),
    "udaseg_conv2d_fwd": (_I, [_D, _P, _P, _P, _P, _I, _F, _I, _P]),
    "udaseg_conv2d_fwd_bnstats": (_I, [_D, _P, _P, _P, _P, _P, _P]),
    "udaseg_conv2d_fwd_fused": (_I, [_D, _P, _P, _P, _P, _P, _I, _F, _P]),
    "udaseg_conv2d_fwd_bf16": (_I, [_D, _P, _P, _P, _P, _P, _I, _I, _F, _P, _P]),
    "udaseg_conv2d_dgrad_bf16": (_I, [_D, _P, _P, _P, _I, _P]),
    "udaseg_conv2d_wgrad_bf16": (_I, [_D, _P, _P, _P, _I, _P]),
    "udaseg_conv2d_dgrad": (_I, [_D, _P, _P, _P, _I, _P]),
    "udaseg_conv2d_wgrad": (_I, [_D, _P, _P, _P, _I, _P]),
    "udaseg_conv2d_dgrad_bnreduce_ok": (_I, [_D]),
    "udaseg_conv2d_dgrad_bnreduce": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P]),
    "udaseg_conv2d_dgrad_bnreduce_bf16_ok": (_I, [_D]),
    "udaseg_conv2d_dgrad_bnreduce_bf16": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P]),
    "udaseg_conv2d_fwd_upcat": (_I, [_D, _P, _P, _I, _P, _P, _P, _I, _F, _P, _P]),
    "udaseg_conv2d_fwd_upcat_bf16": (_I, [_D, _P, _P, _I, _P, _P, _P, _I, _F, _P, _P]),
    "udaseg_conv2d_dgrad_split": (_I, [_D, _P, _P, _P, _P, _I, _P]),
    "udaseg_conv2d_dgrad_split_bf16": (_I, [_D, _P, _P, _P, _P, _I, _P]),
    "udaseg_conv2d_wgrad_part": (_I, [_D, _P, _I, _I, _I, _P, _P, _I, _P]),
    "udaseg_conv2d_wgrad_part_bf16": (_I, [_D, _P, _I, _I, _I, _P, _P, _I, _P]),
    "udaseg_pack_dgrad_weights": (_I, [_D, _P, _P, _P]),
    "udaseg_pack_dgrad_batched": (_I, [_P, _P, _P, _I, _P]),
    "udaseg_conv_flops": (C.c_double, [_D]),
    "udaseg_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "udaseg_bn_replicas": (_I, []),
    "udaseg_bn_stats": (_I, [_P, _L, _I, _P, _P]),
    "udaseg_bn_stats_bf16": (_I, [_P, _L, _I, _P, _P]),
    "udaseg_bn_apply": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _F, _F, _P, _P, _P, _P, _I, _F, _P]),
    "udaseg_bn_apply_eval": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _F, _P]),
    "udaseg_bn_fold": (_I, [_P, _P, _P, _P, _P, _P, _F, _I, _I, _P, _P, _P]),
    "udaseg_bn_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _I, _F, _P]),
    "udaseg_bn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _I, _I, _P]),
    "udaseg_act_bwd": (_I, [_P, _P, _P, _L, _I, _F, _P]),
    "udaseg_channel_sum": (_I, [_P, _L, _I, _P, _I, _P]),
    "udaseg_channel_sum_ws": (_I, [_P, _L, _I, _P, _I, _P, C.c_size_t, _P]),
    "udaseg_channel_sum_scratch_bytes": (C.c_size_t, [_I]),
    "udaseg_maxpool3x3s2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "udaseg_maxpool3x3s2_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "udaseg_upsample2x_concat_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "udaseg_upsample2x_concat_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "udaseg_upsample2x_bilinear_concat_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "udaseg_upsample2x_bilinear_concat_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "udaseg_ce_partials": (_I, []),
    "udaseg_ce_fwd": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P]),
    "udaseg_ce_bwd": (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P]),
    "udaseg_ce_fwd_bwd": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _P]),
    "udaseg_scale_unless_one": (_I, [_P, _L, _P, _I, _P, _P]),
    "udaseg_seg_partials": (_I, []),
    "udaseg_dice_fwd": (_I, [_P, _P, _I, _L, _I, _I, _F, _F, _I, _P, _P, _P, _P]),
    "udaseg_gap_linear_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "udaseg_gap_linear_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "udaseg_bce_logits_target_fwd": (_I, [_P, _P, _I, _F, _P, _I, _P]),
    "udaseg_bce_logits_target_bwd": (_I, [_P, _P, _I, _F, _P, _P, _I, _P]),
    "udaseg_scale_f32": (_I, [_P, _P, _L, _F, _P]),
    "udaseg_prepare_batch_u8": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _P, _I, _P]),
    "udaseg_dice_bwd": (_I, [_P, _P, _P, _P, _F, _I, _L, _I, _I, _P, _I, _P]),
    "udaseg_focal_fwd": (_I, [_P, _P, _P, _F, _F, _L, _I, _I, _I, _P, _P, _I, _P]),
    "udaseg_focal_bwd": (_I, [_P, _P, _P, _F, _F, _P, _F, _L, _I, _I, _P, _I, _P]),
    "udaseg_consistency_fwd": (_I, [_P, _P, _F, _I, _L, _I, _I, _P, _P, _P]),
    "udaseg_consistency_bwd": (_I, [_P, _P, _F, _P, _F, _I, _L, _I, _I, _P, _P, _I, _P]),
    "udaseg_argmax_confusion": (_I, [_P, _P, _L, _I, _I, _P, _P, _P]),
    "udaseg_gap_splits": (_I, [_I]),
    "udaseg_gap_linear_sigmoid_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "udaseg_gap_linear_sigmoid_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "udaseg_bce_logits_fwd": (_I, [_P, _I, _F, _F, _P, _I, _P]),
    "udaseg_bce_logits_bwd": (_I, [_P, _I, _F, _F, _P, _P, _I, _P]),
    "udaseg_adam_flat": (_I, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _F, _P]),
    "udaseg_bn_apply_bf16": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _F, _F, _P, _P, _P, _P, _I, _F, _P]),
    "udaseg_bn_bwd_reduce_bf16": (_I, [_P, _P, _P, _P, _P, _L, _I, _P, _I, _F, _P]),
    "udaseg_bn_bwd_apply_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _I, _I, _P]),
    "udaseg_act_bwd_bf16": (_I, [_P, _P, _P, _L, _I, _F, _P]),
    "udaseg_channel_sum_bf16": (_I, [_P, _L, _I, _P, _I, _P]),
    "udaseg_channel_sum_bf16_ws": (_I, [_P, _L, _I, _P, _I, _P, C.c_size_t, _P]),
    "udaseg_nchw_to_nhwc_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "udaseg_cast_f32_to_bf16": (_I, [_P, _P, _L, _P]),
    "udaseg_maxpool3x3s2_fwd_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "udaseg_maxpool3x3s2_bwd_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "udaseg_upsample2x_concat_bwd_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "udaseg_gap_partial_bf16": (_I, [_P, _P, _I, _I, _I, _P]),
    "udaseg_gap_finish": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "udaseg_gap_bwd_broadcast_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "udaseg_gap_bwd_param": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "udaseg_pack_dgrad_batched_bf16": (_I, [_P, _P, _P, _I, _P]),
    "udaseg_bn_finalize": (_I, [_P, _P, _P, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P]),
    "udaseg_bn_bwd_apply_recompute_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _P]),
    "udaseg_conv2d_wgrad_halo_bf16_ok": (_I, [_D, _I]),
    "udaseg_conv2d_wgrad_halo_bf16": (_I, [_D, _P, _P, _I, _P, _P, _P]),
    "udaseg_conv2d_wgrad_bnin_bf16": (_I, [_D, _P, _P, _P, _I, _F, _P, _P, _I, _P]),
    "udaseg_frag_elems": (_L, [_I, _I, _I]),
    "udaseg_set_stats_scratch": (_I, [_P, C.c_size_t]),
    "udaseg_pack_frag_batched_bf16": (_I, [_P, _P, _P, _P, _I, _P]),
    "udaseg_conv_frag_ok": (_I, [_D, _I, _I]),
    "udaseg_conv_frag_preferred": (_I, [_D, _I, _I]),
    "udaseg_conv2d_fwd_frag_bf16": (_I, [_D, _P, _P, _I, _P, _P, _P, _P, _I, _F, _P, _I, _I, _F, _P, _P]),
    "udaseg_conv2d_dgrad_frag_bf16": (_I, [_D, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _F, _P, _I, _P]),
    "udaseg_pack_frag_batched_f32x3": (_I, [_P, _P, _P, _P, _I, _P]),
    "udaseg_conv2d_wgrad_halo_f32x3_ok": (_I, [_D, _I]),
    "udaseg_conv2d_wgrad_halo_f32x3": (_I, [_D, _P, _P, _I, _P, _P, _P]),
    "udaseg_conv_f32x3_ok": (_I, [_D, _I, _I]),
    "udaseg_f32x3_force_config": (_I, [_I]),
    "udaseg_conv_f32x3_preferred": (_I, [_D, _I, _I]),
    "udaseg_conv2d_fwd_f32x3": (_I, [_D, _P, _P, _I, _P, _P, _P, _I, _F, _P, _P]),
    "udaseg_conv2d_fwd_f32x3_bnin_ok": (_I, [_D, _I]),
    "udaseg_conv2d_fwd_f32x3_bnin_writes": (_I, [_D]),
    "udaseg_conv2d_fwd_f32x3_bnin": (_I, [_D, _P, _I, _P, _P, _I, _F, _P, _P, _P, _P, _I, _F, _P, _P]),
    "udaseg_conv2d_wgrad_bnin_ok": (_I, [_D, _I]),
    "udaseg_conv2d_wgrad_bnin": (_I, [_D, _P, _I, _P, _P, _I, _F, _P, _P, _I, _P]),
    "udaseg_conv2d_dgrad_f32x3": (_I, [_D, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _F, _P, _I, _P]),
    "udaseg_conv_up_f32x3_ok": (_I, [_D, _I]),
    "udaseg_pack_up_batched_f32x3": (_I, [_P, _P, _P, _P, _I, _P]),
    "udaseg_conv2d_fwd_up_f32x3": (_I, [_D, _P, _I, _P, _P, _I, _P, _P]),
    "udaseg_conv2d_dgrad_up_f32x3": (_I, [_D, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _I, _P]),
    "udaseg_up_f32x3_force_config": (_I, [_I]),
    "udaseg_conv2d_wgrad_up_f32x3_ok": (_I, [_D, _I]),
    "udaseg_conv2d_wgrad_up_f32x3": (_I, [_D, _P, _I, _P, _P, _P]),
    "udaseg_conv2d_wgrad_halo_slice_f32x3": (_I, [_D, _P, _P, _P, _I, _I, _P]),
    "udaseg_wgrad_up_set_blocks": (_I, [_I]),
    "udaseg_conv_stem_f32x3_ok": (_I, [_D]),
    "udaseg_conv2d_fwd_stem_f32x3": (_I, [_D, _P, _P, _P, _P, _P]),
    "udaseg_conv_n16_f32x3_ok": (_I, [_D, _I]),
    "udaseg_conv2d_fwd_n16_f32x3": (_I, [_D, _P, _P, _P, _I, _F, _P, _P, _P, _P]),
    "udaseg_conv2d_dgrad_n16_f32x3": (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _P, _P]),
    "udaseg_set_workspace": (_I, [_P, C.c_size_t]),
    "udaseg_workspace_bytes": (C.c_size_t, [_D]),
    "udaseg_debug_set_timeline": (_I, [_P, _I]),
    "udaseg_fill_f32": (_I, [_P, _L, _F, _P]),
    "udaseg_axpy_f32": (_I, [_P, _P, _L, _F, _P]),
    "udaseg_add_i64": (_I, [_P, _L, _L, _P]),
    "udaseg_prof_enable": (_I, [_I]),
    "udaseg_prof_reset": (_I, []),
    "udaseg_prof_read": (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "udaseg_prof_kernel_count": (_I, []),
    "udaseg_prof_kernel_name": (C.c_char_p, [_I]),
    "udaseg_prof_kernel_read": (_I, [_I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "udaseg_prof_records": (_I, [_I, _I, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}

# The library's switchboard (include/udaseg.h UDASEG_OPT_*, table in csrc/api.hip): name -> key.  The ONE Python mirror of it;
# tests/test_abi.py checks it against udaseg_option_count / udaseg_option_name.  kernels.set_option / get_option take these names.
OPTIONS = {"GENERIC_GATHER": 0, "F32_SPLIT": 1, "WGRAD_GENERIC": 2, "F32_HALO": 3, "F3_CFG": 4, "F3_WS": 5, "F3_SIGNS": 6, "IGEMM_TILE": 7, "IGEMM_X3": 8, "NO_FOLD": 9, "WGRAD_X3_BLOCKS": 10, "WGRAD_BLOCKS": 11, "WGRAD_NO_XCD": 12, "WGRAD_X3": 13, "NO_WGRAD_HALO": 14, "WGRAD_F3_BLOCKS": 15, "WGRAD_HALO_BLOCKS": 16, "WGRAD_DEEP_BLOCKS": 17, "WGRAD_DB": 18, "REDUCE_BLOCKS": 19, "BN_APPLY_PT": 20, "GEMM_1X1_TILE": 21, "GEMM_1X1": 22, "GEMM_1X1_MAXM": 23, "NO_STREAM": 24, "HALO_CFG": 25, "NO_HALO": 26, "NO_HALO_S2": 27, "HALO_W16": 28, "HALO_DEEP": 29, "HALO_S2_CK": 30, "UP_CFG": 31, "WGRAD_UP_BLOCKS": 32}

_lib = None


def load():
    """Load the shared library once; raise loudly when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c \"import __graft_entry__ as g; "
            f"g.build()\"` (or `make -C {os.path.join(_HERE, 'csrc')}`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().udaseg_last_error().decode(errors="replace")
        raise RuntimeError(f"libudaseg_hip {what} failed (rc={rc}): {msg}")


def require_gpu():
    lib = load()
    if lib.udaseg_device_count() < 1:
        raise RuntimeError("libudaseg_hip: no HIP device visible; this path has no CPU fallback")
    return lib
