"""The ONE table every pointer-passing call into libudaseg_hip.so goes through: entry point -> role of each C argument.

The C-ABI takes plain pointers and ONE set of extents per call (include/udaseg.h): a tensor of another dtype, a short buffer or
a tensor on another device is read / written past its end by the kernel -- a GPU memory fault, not an error code (rounds 2 and 3
each had one: profiles/r02_bn_bandwidth.txt, profiles/r03_gemm_1x1.txt; both times the binding had dispatched on ONE tensor's
dtype and passed every other operand as a raw pointer).  Here every tensor operand of every entry point declares

    T(name, dtype, count[, opt])    dtype: what the kernel reads / writes the bytes as; count: the fewest elements it touches,
                                    an expression over the call's scalar arguments and the conv descriptor's fields

and ``ops.<entry point>(*args)`` -- generated from the table at import -- checks dtype, element count, contiguity, device (all
tensors of a call on ONE GPU) and NULL-ness BEFORE any ``data_ptr()`` is taken; a violation raises ``ValueError`` and the library
is not called.  ``tests/test_abi.py`` walks every name of ``_lib.SIGNATURES``: each has a row here whose argument kinds match the
ctypes signature, and for each tensor role a wrong dtype and a short buffer must raise before the (stubbed) library is reached.

Names usable in count / dtype expressions: the scalar arguments of the call, the descriptor fields n hi wi ci ho wo co kh kw
stride pad, X = n*hi*wi*ci, Y = n*ho*wo*co, W = co*kh*kw*ci, R (udaseg_bn_replicas), frag(n_out, k_in, ks), gap_splits(hw),
ce_partials(), seg_partials(), and the dtypes f32 bf16 f64 i64 i32 u8.  dtype "raw32": any dtype, count in 4-byte units (pure
data movement in 16-byte vectors: udaseg_upsample2x_concat_fwd moves bf16 tensors as half as many fp32 "channels").
"""
import ctypes as C

import torch

from . import _lib

f32, bf16, f64, i64, i32, u8 = torch.float32, torch.bfloat16, torch.float64, torch.int64, torch.int32, torch.uint8
_DT = {"f32": f32, "bf16": bf16, "f64": f64, "i64": i64, "i32": i32, "u8": u8}

D = ("desc",)            # const udaseg_conv_desc*
S = ("stream",)          # void* stream (None -> torch's current stream)


def I(name):             # int / int64_t / size_t scalar
    return ("int", name)


def F(name):             # float scalar
    return ("float", name)


def H(name):             # host pointer (ctypes array / byref), passed through
    return ("host", name)


def T(name, dtype, count, opt=False):
    return ("tensor", name, dtype, str(count), opt)


_HALF = "n*(hi//2)*(wi//2)"          # pixels of the half-resolution source of a fused decoder input
_ACT_BF = "(f32 if out_f32 else bf16)"
_BIL = "(bf16 if bf16_ else f32)"

OPERANDS = {
    # ---- library state / queries (no device pointers)
    "udaseg_version": [], "udaseg_last_error": [], "udaseg_device_count": [], "udaseg_bn_replicas": [], "udaseg_ce_partials": [],
    "udaseg_seg_partials": [], "udaseg_prof_reset": [], "udaseg_prof_kernel_count": [],
    "udaseg_set_option": [I("key"), I("value")], "udaseg_get_option": [I("key")], "udaseg_option_count": [],
    "udaseg_option_name": [I("key")], "udaseg_option_epoch": [],
    "udaseg_conv2d_dgrad_bnreduce_ok": [D], "udaseg_conv2d_dgrad_bnreduce_bf16_ok": [D], "udaseg_conv_flops": [D],
    "udaseg_conv2d_fwd_f32x3_bnin_ok": [D, I("up")], "udaseg_conv2d_wgrad_bnin_ok": [D, I("up")],
    "udaseg_conv2d_fwd_f32x3_bnin_writes": [D],
    "udaseg_workspace_bytes": [D],
    "udaseg_channel_sum_scratch_bytes": [I("c")], "udaseg_gap_splits": [I("hw")], "udaseg_frag_elems": [I("n_out"), I("k_in"), I("ks")],
    "udaseg_conv2d_wgrad_halo_bf16_ok": [D, I("up_ca")], "udaseg_conv2d_wgrad_halo_f32x3_ok": [D, I("up_ca")],
    "udaseg_conv_frag_ok": [D, I("dgrad"), I("up_ca")], "udaseg_conv_frag_preferred": [D, I("dgrad"), I("up_ca")],
    "udaseg_conv_f32x3_ok": [D, I("dgrad"), I("up_ca")], "udaseg_conv_f32x3_preferred": [D, I("dgrad"), I("up_ca")],
    "udaseg_f32x3_force_config": [I("cfg")], "udaseg_up_f32x3_force_config": [I("cfg")],
    "udaseg_conv_up_f32x3_ok": [D, I("up_ca")], "udaseg_conv2d_wgrad_up_f32x3_ok": [D, I("up_ca")],
    "udaseg_wgrad_up_set_blocks": [I("blocks")], "udaseg_conv_n16_f32x3_ok": [D, I("dgrad")],
    "udaseg_conv_stem_f32x3_ok": [D],
    "udaseg_prof_enable": [I("on")], "udaseg_prof_kernel_name": [I("kid")],
    "udaseg_prof_read": [I("family"), H("total_ms"), H("total_flops"), H("launches")],
    "udaseg_prof_kernel_read": [I("kid"), H("total_ms"), H("total_flops"), H("launches")],
    "udaseg_prof_records": [I("family"), I("max_records"), H("ms"), H("flops"), H("kind"), H("desc11")],
    # caller-owned scratch bound to the current device
    "udaseg_set_workspace": [T("ptr", u8, "bytes", opt=True), I("bytes")],
    "udaseg_memset_async": [T("ptr", "raw32", "(bytes+3)//4"), I("value"), I("bytes"), S],
    "udaseg_stream_wait": [H("waiter"), H("signal")],      # two hipStream_t handles, passed through
    "udaseg_set_stats_scratch": [T("ptr", u8, "bytes", opt=True), I("bytes")],
    "udaseg_debug_set_timeline": [T("buffer", i64, "6*blocks", opt=True), I("blocks")],
    # ---- convolutions, fp32 storage
    "udaseg_conv2d_fwd": [D, T("x", f32, "X"), T("w", f32, "W"), T("bias", f32, "co", True), T("y", f32, "Y"), I("act"), F("slope"),
                          I("accumulate"), S],
    "udaseg_conv2d_fwd_bnstats": [D, T("x", f32, "X"), T("w", f32, "W"), T("bias", f32, "co", True), T("y", f32, "Y"),
                                  T("stats", f64, "2*co*R"), S],
    "udaseg_conv2d_fwd_fused": [D, T("x", f32, "X"), T("w", f32, "W"), T("bias", f32, "co", True), T("residual", f32, "Y", True),
                                T("y", f32, "Y"), I("act"), F("slope"), S],
    "udaseg_conv2d_dgrad": [D, T("dy", f32, "Y"), T("w_t", f32, "W"), T("dx", f32, "X"), I("accumulate"), S],
    "udaseg_conv2d_dgrad_bnreduce": [D, T("dy", f32, "Y"), T("w_t", f32, "W"), T("dx", f32, "X"), T("prev_y", f32, "X"),
                                     T("save_mean", f32, "ci"), T("save_rstd", f32, "ci"), T("gamma", f32, "ci"), T("beta", f32, "ci"),
                                     I("act"), F("slope"), T("bsums", f64, "2*ci*R"), S],
    "udaseg_conv2d_wgrad": [D, T("x", f32, "X"), T("dy", f32, "Y"), T("dw", f32, "W"), I("accumulate"), S],
    "udaseg_conv2d_fwd_upcat": [D, T("a", f32, _HALF + "*ca"), T("skip", f32, "n*hi*wi*(ci-ca)", True), I("ca"), T("w", f32, "W"),
                                T("bias", f32, "co", True), T("y", f32, "Y"), I("act"), F("slope"), T("stats", f64, "2*co*R", True), S],
    "udaseg_conv2d_dgrad_split": [D, T("dy", f32, "Y"), T("w_t", f32, "W"), T("dx_a", f32, "n*hi*wi*ca"),
                                  T("dx_b", f32, "n*hi*wi*(ci-ca)"), I("ca"), S],
    "udaseg_conv2d_wgrad_part": [D, T("src", f32, "(" + _HALF + " if up else n*hi*wi)*src_c"), I("src_c"), I("c_off"), I("up"),
                                 T("dy", f32, "Y"), T("dw", f32, "W"), I("accumulate"), S],
    "udaseg_pack_dgrad_weights": [D, T("w", f32, "W"), T("w_t", f32, "W"), S],
    "udaseg_pack_dgrad_batched": [T("arena", f32, 1), T("packed", f32, 1), T("table", i32, "5*entries"), I("entries"), S],
    "udaseg_conv2d_fwd_f32x3": [D, T("x", f32, "(" + _HALF + "*up_ca if up_ca else X)"), T("skip", f32, "n*hi*wi*(ci-up_ca)", True),
                                I("up_ca"), T("wfrag3", bf16, "3*frag(co,ci,kh)"), T("bias", f32, "co", True), T("y", f32, "Y"),
                                I("act"), F("slope"), T("stats", f64, "2*co*R", True), S],
    "udaseg_conv2d_fwd_f32x3_bnin": [D, T("x", f32, "(" + _HALF + "*ci if up else X)"), I("up"), T("in_scale", f32, "ci"),
                                     T("in_shift", f32, "ci"), I("in_act"),
                                     F("in_slope"), T("z_out", f32, "X", True), T("wfrag3", bf16, "3*frag(co,ci,kh)"),
                                     T("bias", f32, "co", True), T("y", f32, "Y"),
                                     I("act"), F("slope"), T("stats", f64, "2*co*R", True), S],
    "udaseg_conv2d_wgrad_bnin": [D, T("x", f32, "(" + _HALF + "*ci if up else X)"), I("up"), T("in_scale", f32, "ci"),
                                 T("in_shift", f32, "ci"), I("in_act"),
                                 F("in_slope"), T("dy", f32, "Y"), T("dw", f32, "W"), I("accumulate"), S],
    "udaseg_conv2d_dgrad_f32x3": [D, T("dy", f32, "Y"), T("wfrag3_t", bf16, "3*frag(ci,co,kh)"),
                                  T("dx", f32, "n*hi*wi*(split if split else ci)"), T("dx2", f32, "n*hi*wi*(ci-split)", True),
                                  I("split"), T("prev_y", f32, "X", True), T("save_mean", f32, "ci", True),
                                  T("save_rstd", f32, "ci", True), T("gamma", f32, "ci", True), T("beta", f32, "ci", True),
                                  I("bn_act"), F("bn_slope"), T("bsums", f64, "2*ci*R", True), I("accumulate"), S],
    "udaseg_conv2d_wgrad_halo_f32x3": [D, T("x", f32, "(" + _HALF + "*up_ca if up_ca else X)"),
                                       T("skip", f32, "n*hi*wi*(ci-up_ca)", True), I("up_ca"), T("dy", f32, "Y"), T("dw", f32, "W"), S],
    "udaseg_pack_up_batched_f32x3": [T("w32", f32, 1, True), T("wt32", f32, 1, True), T("packed", bf16, 1),
                                     T("table", i32, "8*entries"), I("entries"), S],
    "udaseg_conv2d_fwd_up_f32x3": [D, T("a", f32, _HALF + "*up_c"), I("up_c"), T("wfrag_up", bf16, "3*frag(co,up_c,4)"),
                                   T("y", f32, "Y"), I("accumulate"), T("stats", f64, "2*co*R", True), S],
    "udaseg_conv2d_dgrad_up_f32x3": [D, T("dy", f32, "Y"), I("up_c"), T("wfrag_up_t", bf16, "3*frag(up_c,co,4)"),
                                     T("da", f32, _HALF + "*up_c"), T("prev_y", f32, _HALF + "*up_c", True),
                                     T("save_mean", f32, "up_c", True), T("save_rstd", f32, "up_c", True), T("gamma", f32, "up_c", True),
                                     T("beta", f32, "up_c", True), I("bn_act"), F("bn_slope"), T("bsums", f64, "2*up_c*R", True),
                                     I("accumulate"), S],
    "udaseg_conv2d_wgrad_up_f32x3": [D, T("a", f32, _HALF + "*up_c"), I("up_c"), T("dy", f32, "Y"), T("dw", f32, "W"), S],
    "udaseg_conv2d_wgrad_halo_slice_f32x3": [D, T("x", f32, "X"), T("dy", f32, "Y"), T("dw", f32, "co*kh*kw*ldw_"), I("ldw_"),
                                             I("c_off"), S],
    "udaseg_conv2d_fwd_stem_f32x3": [D, T("x", f32, "X"), T("wfrag", bf16, "3*28*512"), T("y", f32, "Y"), T("stats", f64, "2*co*R", True), S],
    "udaseg_conv2d_fwd_n16_f32x3": [D, T("x", f32, "X"), T("in_scale", f32, "ci", True), T("in_shift", f32, "ci", True), I("in_act"),
                                    F("in_slope"), T("wfrag", bf16, "3*((ci+15)//16)*5*512"), T("y", f32, "Y"),
                                    T("stats", f64, "2*co*R", True), S],
    "udaseg_conv2d_dgrad_n16_f32x3": [D, T("dy", f32, "Y"), T("wfrag_t", bf16, "3*((co+15)//16)*5*512"), T("dx", f32, "X"),
                                      T("prev_y", f32, "X", True), T("save_mean", f32, "ci", True), T("save_rstd", f32, "ci", True),
                                      T("gamma", f32, "ci", True), T("beta", f32, "ci", True), I("bn_act"), F("bn_slope"),
                                      T("bsums", f64, "2*ci*R", True), S],
    "udaseg_pack_frag_batched_f32x3": [T("w32", f32, 1, True), T("wt32", f32, 1, True), T("packed", bf16, 1),
                                       T("table", i32, "6*entries"), I("entries"), S],
    # ---- convolutions, bf16 storage
    "udaseg_conv2d_fwd_bf16": [D, T("x", bf16, "X"), T("w", bf16, "W"), T("bias", f32, "co", True), T("residual", bf16, "Y", True),
                               T("y", _ACT_BF, "Y"), I("out_f32"), I("act"), F("slope"), T("stats", f64, "2*co*R", True), S],
    "udaseg_conv2d_dgrad_bf16": [D, T("dy", bf16, "Y"), T("w_t", bf16, "W"), T("dx", bf16, "X"), I("accumulate"), S],
    "udaseg_conv2d_wgrad_bf16": [D, T("x", bf16, "X"), T("dy", bf16, "Y"), T("dw", f32, "W"), I("accumulate"), S],
    "udaseg_conv2d_dgrad_bnreduce_bf16": [D, T("dy", bf16, "Y"), T("w_t", bf16, "W"), T("dx", bf16, "X"), T("prev_y", bf16, "X"),
                                          T("save_mean", f32, "ci"), T("save_rstd", f32, "ci"), T("gamma", f32, "ci"),
                                          T("beta", f32, "ci"), I("act"), F("slope"), T("bsums", f64, "2*ci*R"), S],
    "udaseg_conv2d_fwd_upcat_bf16": [D, T("a", bf16, _HALF + "*ca"), T("skip", bf16, "n*hi*wi*(ci-ca)", True), I("ca"),
                                     T("w", bf16, "W"), T("bias", f32, "co", True), T("y", bf16, "Y"), I("act"), F("slope"),
                                     T("stats", f64, "2*co*R", True), S],
    "udaseg_conv2d_dgrad_split_bf16": [D, T("dy", bf16, "Y"), T("w_t", bf16, "W"), T("dx_a", bf16, "n*hi*wi*ca"),
                                       T("dx_b", bf16, "n*hi*wi*(ci-ca)"), I("ca"), S],
    "udaseg_conv2d_wgrad_part_bf16": [D, T("src", bf16, "(" + _HALF + " if up else n*hi*wi)*src_c"), I("src_c"), I("c_off"), I("up"),
                                      T("dy", bf16, "Y"), T("dw", f32, "W"), I("accumulate"), S],
    "udaseg_pack_dgrad_batched_bf16": [T("arena", f32, 1), T("packed", bf16, 1), T("table", i32, "5*entries"), I("entries"), S],
    "udaseg_conv2d_wgrad_halo_bf16": [D, T("x", bf16, "(" + _HALF + "*up_ca if up_ca else X)"),
                                      T("skip", bf16, "n*hi*wi*(ci-up_ca)", True), I("up_ca"), T("dy", bf16, "Y"), T("dw", f32, "W"), S],
    "udaseg_conv2d_wgrad_bnin_bf16": [D, T("x", bf16, "X"), T("in_scale", f32, "ci"), T("in_shift", f32, "ci"), I("in_act"),
                                      F("in_slope"), T("dy", bf16, "Y"), T("dw", f32, "W"), I("accumulate"), S],
    "udaseg_pack_frag_batched_bf16": [T("w16", bf16, 1, True), T("wt16", bf16, 1, True), T("packed", bf16, 1),
                                      T("table", i32, "6*entries"), I("entries"), S],
    "udaseg_conv2d_fwd_frag_bf16": [D, T("x", bf16, "(" + _HALF + "*up_ca if up_ca else X)"),
                                    T("skip", bf16, "n*hi*wi*(ci-up_ca)", True), I("up_ca"), T("wfrag", bf16, "(frag(co,4*ci,2) if kh == 4 else frag(co,ci,kh))"),
                                    T("bias", f32, "co", True), T("in_scale", f32, "ci", True), T("in_shift", f32, "ci", True),
                                    I("in_act"), F("in_slope"), T("y", _ACT_BF, "Y"), I("out_f32"), I("act"), F("slope"),
                                    T("stats", f64, "2*co*R", True), S],
    "udaseg_conv2d_dgrad_frag_bf16": [D, T("dy", bf16, "Y"), T("wfrag_t", bf16, "(4*frag(ci,co,2) if kh == 4 else frag(ci,co,kh))"),
                                      T("dx", bf16, "n*hi*wi*(split if split else ci)"), T("dx2", bf16, "n*hi*wi*(ci-split)", True),
                                      I("split"), T("prev_y", bf16, "X", True), T("save_mean", f32, "ci", True),
                                      T("save_rstd", f32, "ci", True), T("gamma", f32, "ci", True), T("beta", f32, "ci", True),
                                      I("bn_act"), F("bn_slope"), T("bsums", f64, "2*ci*R", True), I("accumulate"), S],
    # ---- layout / casts
    "udaseg_nchw_to_nhwc": [T("x", f32, "n*c*h*w"), T("y", f32, "n*h*w*cpad"), I("n"), I("c"), I("h"), I("w"), I("cpad"), S],
    "udaseg_nchw_to_nhwc_bf16": [T("x", f32, "n*c*h*w"), T("y", bf16, "n*h*w*cpad"), I("n"), I("c"), I("h"), I("w"), I("cpad"), S],
    "udaseg_cast_f32_to_bf16": [T("x", f32, "count"), T("y", bf16, "count"), I("count"), S],
    "udaseg_prepare_batch_u8": [T("images", u8, "n*h*w*3"), T("masks", u8, "n*h*w", True), T("d4", i32, "n", True), I("n"), I("h"),
                                I("w"), H("mean255"), H("inv_std255"), T("out_images", "(bf16 if out_bf16 else f32)", "n*h*w*cpad"),
                                I("cpad"), I("out_bf16"), T("out_masks", i64, "n*h*w", True), I("square_checked"), S],
    # ---- BatchNorm / activation passes
    "udaseg_bn_stats": [T("y", f32, "pixels*c"), I("pixels"), I("c"), T("sums", f64, "2*c*R"), S],
    "udaseg_bn_stats_bf16": [T("y", bf16, "pixels*c"), I("pixels"), I("c"), T("sums", f64, "2*c*R"), S],
    "udaseg_bn_apply": [T("y", f32, "pixels*c"), T("sums", f64, "2*c*R"), T("gamma", f32, "c"), T("beta", f32, "c"),
                        T("residual", f32, "pixels*c", True), T("z", f32, "pixels*c"), I("pixels"), I("c"), F("eps"), F("momentum"),
                        T("running_mean", f32, "c", True), T("running_var", f32, "c", True), T("save_mean", f32, "c", True),
                        T("save_rstd", f32, "c", True), I("act"), F("slope"), S],
    "udaseg_bn_apply_bf16": [T("y", bf16, "pixels*c"), T("sums", f64, "2*c*R"), T("gamma", f32, "c"), T("beta", f32, "c"),
                             T("residual", bf16, "pixels*c", True), T("z", bf16, "pixels*c"), I("pixels"), I("c"), F("eps"),
                             F("momentum"), T("running_mean", f32, "c", True), T("running_var", f32, "c", True),
                             T("save_mean", f32, "c", True), T("save_rstd", f32, "c", True), I("act"), F("slope"), S],
    "udaseg_bn_apply_eval": [T("y", f32, "pixels*c"), T("gamma", f32, "c"), T("beta", f32, "c"), T("running_mean", f32, "c"),
                             T("running_var", f32, "c"), T("residual", f32, "pixels*c", True), T("z", f32, "pixels*c"), I("pixels"),
                             I("c"), F("eps"), I("act"), F("slope"), S],
    "udaseg_bn_fold": [T("w", f32, "co*row_len"), T("bias", f32, "co", True), T("gamma", f32, "co"), T("beta", f32, "co"),
                       T("running_mean", f32, "co"), T("running_var", f32, "co"), F("eps"), I("co"), I("row_len"),
                       T("w_folded", f32, "co*row_len"), T("bias_folded", f32, "co"), S],
    "udaseg_bn_finalize": [T("sums", f64, "2*c*R"), T("gamma", f32, "c"), T("beta", f32, "c"), I("pixels"), I("c"), F("eps"),
                           F("momentum"), T("running_mean", f32, "c", True), T("running_var", f32, "c", True),
                           T("save_mean", f32, "c", True), T("save_rstd", f32, "c", True), T("scale", f32, "c"), T("shift", f32, "c"), S],
    "udaseg_bn_bwd_reduce": [T("dz", f32, "pixels*c"), T("z", f32, "pixels*c", True), T("y", f32, "pixels*c"), T("save_mean", f32, "c"),
                             T("save_rstd", f32, "c"), T("gamma", f32, "c", True), T("beta", f32, "c", True), I("pixels"), I("c"),
                             T("bsums", f64, "2*c*R"), I("act"), F("slope"), S],
    "udaseg_bn_bwd_reduce_bf16": [T("dz", bf16, "pixels*c"), T("z", bf16, "pixels*c", True), T("y", bf16, "pixels*c"),
                                  T("save_mean", f32, "c"), T("save_rstd", f32, "c"), I("pixels"), I("c"), T("bsums", f64, "2*c*R"),
                                  I("act"), F("slope"), S],
    "udaseg_bn_bwd_apply": [T("dz", f32, "pixels*c"), T("z", f32, "pixels*c", True), T("y", f32, "pixels*c"), T("save_mean", f32, "c"),
                            T("save_rstd", f32, "c"), T("gamma", f32, "c"), T("beta", f32, "c", True), T("bsums", f64, "2*c*R"),
                            T("dy", f32, "pixels*c"), T("dres", f32, "pixels*c", True), T("dgamma", f32, "c", True),
                            T("dbeta", f32, "c", True), I("pixels"), I("c"), I("act"), F("slope"), I("accumulate_dy"),
                            I("accumulate_dres"), I("accumulate_param"), S],
    "udaseg_bn_bwd_apply_bf16": [T("dz", bf16, "pixels*c"), T("z", bf16, "pixels*c", True), T("y", bf16, "pixels*c"),
                                 T("save_mean", f32, "c"), T("save_rstd", f32, "c"), T("gamma", f32, "c"), T("bsums", f64, "2*c*R"),
                                 T("dy", bf16, "pixels*c"), T("dres", bf16, "pixels*c", True), T("dgamma", f32, "c", True),
                                 T("dbeta", f32, "c", True), I("pixels"), I("c"), I("act"), F("slope"), I("accumulate_dy"),
                                 I("accumulate_dres"), I("accumulate_param"), S],
    "udaseg_bn_bwd_apply_recompute_bf16": [T("dz", bf16, "pixels*c"), T("y", bf16, "pixels*c"), T("fwd_scale", f32, "c"),
                                           T("fwd_shift", f32, "c"), T("save_mean", f32, "c"), T("save_rstd", f32, "c"),
                                           T("gamma", f32, "c"), T("bsums", f64, "2*c*R"), T("dy", bf16, "pixels*c"),
                                           T("dgamma", f32, "c", True), T("dbeta", f32, "c", True), I("pixels"), I("c"), I("act"),
                                           F("slope"), S],
    "udaseg_act_bwd": [T("dz", f32, "count"), T("z", f32, "count"), T("dy", f32, "count"), I("count"), I("act"), F("slope"), S],
    "udaseg_act_bwd_bf16": [T("dz", bf16, "count"), T("z", bf16, "count"), T("dy", bf16, "count"), I("count"), I("act"), F("slope"), S],
    "udaseg_channel_sum": [T("x", f32, "pixels*c"), I("pixels"), I("c"), T("out", f32, "c"), I("accumulate"), S],
    "udaseg_channel_sum_ws": [T("x", f32, "pixels*c"), I("pixels"), I("c"), T("out", f32, "c"), I("accumulate"),
                              T("scratch", f32, "scratch_bytes//4"), I("scratch_bytes"), S],
    "udaseg_channel_sum_bf16": [T("x", bf16, "pixels*c"), I("pixels"), I("c"), T("out", f32, "c"), I("accumulate"), S],
    "udaseg_channel_sum_bf16_ws": [T("x", bf16, "pixels*c"), I("pixels"), I("c"), T("out", f32, "c"), I("accumulate"),
                                   T("scratch", f32, "scratch_bytes//4"), I("scratch_bytes"), S],
    # ---- pooling / resize
    "udaseg_maxpool3x3s2_fwd": [T("x", f32, "n*h*w*c"), T("y", f32, "n*((h-1)//2+1)*((w-1)//2+1)*c"),
                                T("idx", u8, "n*((h-1)//2+1)*((w-1)//2+1)*c"), I("n"), I("h"), I("w"), I("c"), S],
    "udaseg_maxpool3x3s2_fwd_bf16": [T("x", bf16, "n*h*w*c"), T("y", bf16, "n*((h-1)//2+1)*((w-1)//2+1)*c"),
                                     T("idx", u8, "n*((h-1)//2+1)*((w-1)//2+1)*c"), I("n"), I("h"), I("w"), I("c"), S],
    "udaseg_maxpool3x3s2_bwd": [T("dy", f32, "n*((h-1)//2+1)*((w-1)//2+1)*c"), T("idx", u8, "n*((h-1)//2+1)*((w-1)//2+1)*c"),
                                T("dx", f32, "n*h*w*c"), I("n"), I("h"), I("w"), I("c"), I("accumulate"), S],
    "udaseg_maxpool3x3s2_bwd_bf16": [T("dy", bf16, "n*((h-1)//2+1)*((w-1)//2+1)*c"), T("idx", u8, "n*((h-1)//2+1)*((w-1)//2+1)*c"),
                                     T("dx", bf16, "n*h*w*c"), I("n"), I("h"), I("w"), I("c"), I("accumulate"), S],
    "udaseg_upsample2x_concat_fwd": [T("a", "raw32", "n*h*w*ca"), T("skip", "raw32", "4*n*h*w*cb", True),
                                     T("out", "raw32", "4*n*h*w*(ca+cb)"), I("n"), I("h"), I("w"), I("ca"), I("cb"), S],
    "udaseg_upsample2x_concat_bwd": [T("dout", f32, "4*n*h*w*(ca+cb)"), T("da", f32, "n*h*w*ca", True),
                                     T("dskip", f32, "4*n*h*w*cb", True), I("n"), I("h"), I("w"), I("ca"), I("cb"), I("accumulate_da"),
                                     I("accumulate_dskip"), S],
    "udaseg_upsample2x_concat_bwd_bf16": [T("dout", bf16, "4*n*h*w*(ca+cb)"), T("da", bf16, "n*h*w*ca", True),
                                          T("dskip", bf16, "4*n*h*w*cb", True), I("n"), I("h"), I("w"), I("ca"), I("cb"),
                                          I("accumulate_da"), I("accumulate_dskip"), S],
    "udaseg_upsample2x_bilinear_concat_fwd": [T("a", _BIL, "n*h*w*ca"), T("skip", _BIL, "4*n*h*w*cb", True),
                                              T("out", _BIL, "4*n*h*w*(ca+cb)"), I("n"), I("h"), I("w"), I("ca"), I("cb"), I("bf16_"), S],
    "udaseg_upsample2x_bilinear_concat_bwd": [T("dout", _BIL, "4*n*h*w*(ca+cb)"), T("da", _BIL, "n*h*w*ca", True),
                                              T("dskip", _BIL, "4*n*h*w*cb", True), I("n"), I("h"), I("w"), I("ca"), I("cb"),
                                              I("accumulate_da"), I("accumulate_dskip"), I("bf16_"), S],
    # ---- losses / metrics (logits: NHWC rows of ldc floats)
    "udaseg_ce_fwd": [T("logits", f32, "pixels*ldc"), T("target", i64, "pixels"), I("pixels"), I("classes"), I("ldc"),
                      T("lse", f32, "pixels"), T("partials", f64, "ce_partials()"), T("loss", f32, 1), S],
    "udaseg_ce_bwd": [T("logits", f32, "pixels*ldc"), T("target", i64, "pixels"), T("lse", f32, "pixels"), T("grad_out", f32, 1, True),
                      I("pixels"), I("classes"), I("ldc"), T("dlogits", f32, "pixels*ldc"),
                      T("colsum_partials", f32, "ce_partials()*ldc", True), T("colsum", f32, "ldc", True), S],
    "udaseg_ce_fwd_bwd": [T("logits", f32, "pixels*ldc"), T("target", i64, "pixels"), I("pixels"), I("classes"), I("ldc"),
                          T("partials", f64, "ce_partials()"), T("loss", f32, 1), T("dlogits", f32, "pixels*ldc"),
                          T("colsum_partials", f32, "ce_partials()*ldc", True), T("colsum", f32, "ldc", True), S],
    "udaseg_scale_unless_one": [T("x", f32, "count"), I("count"), T("x2", f32, "count2", True), I("count2"), T("g", f32, 1), S],
    "udaseg_argmax_confusion": [T("logits", f32, "pixels*ldc"), T("target", i64, "pixels"), I("pixels"), I("classes"), I("ldc"),
                                T("confusion", i64, "classes*classes"), T("pred", i64, "pixels", True), S],
    "udaseg_dice_fwd": [T("logits", f32, "batch*pix_per_image*ldc"), T("target", i64, "batch*pix_per_image"), I("batch"),
                        I("pix_per_image"), I("classes"), I("ldc"), F("smooth"), F("eps"), I("pooled"),
                        T("sums", f64, "batch*3*classes"), T("coef", f32, "batch*2*classes"), T("loss", f32, 1), S],
    "udaseg_dice_bwd": [T("logits", f32, "batch*pix_per_image*ldc"), T("target", i64, "batch*pix_per_image"),
                        T("coef", f32, "batch*2*classes"), T("grad_out", f32, 1, True), F("weight"), I("batch"), I("pix_per_image"),
                        I("classes"), I("ldc"), T("dlogits", f32, "batch*pix_per_image*ldc"), I("accumulate"), S],
    "udaseg_focal_fwd": [T("logits", f32, "pixels*ldc"), T("target", i64, "pixels"), T("class_weights", f32, "classes", True),
                         F("alpha"), F("gamma"), I("pixels"), I("classes"), I("ldc"), I("mean"), T("partials", f64, "seg_partials()"),
                         T("loss", f32, 1), I("accumulate"), S],
    "udaseg_focal_bwd": [T("logits", f32, "pixels*ldc"), T("target", i64, "pixels"), T("class_weights", f32, "classes", True),
                         F("alpha"), F("gamma"), T("grad_out", f32, 1, True), F("weight"), I("pixels"), I("classes"), I("ldc"),
                         T("dlogits", f32, "pixels*ldc"), I("accumulate"), S],
    "udaseg_consistency_fwd": [T("z1", f32, "pixels*ldc"), T("z2", f32, "pixels*ldc"), F("temperature"), I("batch"),
                               I("pixels"), I("classes"), I("ldc"), T("partials", f64, "seg_partials()"), T("loss", f32, 1), S],
    "udaseg_consistency_bwd": [T("z1", f32, "pixels*ldc"), T("z2", f32, "pixels*ldc"), F("temperature"),
                               T("grad_out", f32, 1, True), F("weight"), I("batch"), I("pixels"), I("classes"), I("ldc"),
                               T("d1", f32, "pixels*ldc", True), T("d2", f32, "pixels*ldc", True), I("accumulate"), S],
    # ---- discriminator tail
    "udaseg_gap_linear_sigmoid_fwd": [T("z", f32, "n*hw*c"), T("w", f32, "c"), T("b", f32, 1), T("partial", f32, "n*gap_splits(hw)*c"),
                                      T("pooled", f32, "n*c"), T("p", f32, "n"), I("n"), I("hw"), I("c"), S],
    "udaseg_gap_linear_sigmoid_bwd": [T("dp", f32, "n"), T("p", f32, "n"), T("pooled", f32, "n*c"), T("w", f32, "c"),
                                      T("dz", f32, "n*hw*c"), T("dw", f32, "c"), T("db", f32, 1), I("n"), I("hw"), I("c"),
                                      I("accumulate_param"), S],
    "udaseg_gap_linear_fwd": [T("z", f32, "n*hw*c"), T("w", f32, "c"), T("b", f32, 1), T("partial", f32, "n*gap_splits(hw)*c"),
                              T("pooled", f32, "n*c"), T("logit", f32, "n"), I("n"), I("hw"), I("c"), S],
    "udaseg_gap_linear_bwd": [T("dlogit", f32, "n"), T("pooled", f32, "n*c"), T("w", f32, "c"), T("dz", f32, "n*hw*c"),
                              T("dw", f32, "c"), T("db", f32, 1), I("n"), I("hw"), I("c"), I("accumulate_param"), S],
    "udaseg_gap_partial_bf16": [T("z", bf16, "n*hw*c"), T("partial", f32, "n*gap_splits(hw)*c"), I("n"), I("hw"), I("c"), S],
    "udaseg_gap_finish": [T("partial", f32, "n*gap_splits(hw)*c"), T("w", f32, "c"), T("b", f32, 1), T("pooled", f32, "n*c"),
                          T("p", f32, "n"), I("n"), I("hw"), I("c"), S],
    "udaseg_gap_bwd_broadcast_bf16": [T("dp", f32, "n"), T("p", f32, "n"), T("w", f32, "c"), T("dz", bf16, "n*hw*c"), I("n"), I("hw"),
                                      I("c"), S],
    "udaseg_gap_bwd_param": [T("dp", f32, "n"), T("p", f32, "n"), T("pooled", f32, "n*c"), T("dw", f32, "c"), T("db", f32, 1), I("n"),
                             I("c"), I("accumulate_param"), S],
    "udaseg_bce_logits_fwd": [T("x", f32, "n"), I("n"), F("label"), F("weight"), T("loss", f32, 1), I("accumulate"), S],
    "udaseg_bce_logits_bwd": [T("x", f32, "n"), I("n"), F("label"), F("weight"), T("grad_out", f32, 1, True), T("dx", f32, "n"),
                              I("accumulate"), S],
    "udaseg_bce_logits_target_fwd": [T("x", f32, "n"), T("target", f32, "n"), I("n"), F("weight"), T("loss", f32, 1), I("accumulate"), S],
    "udaseg_bce_logits_target_bwd": [T("x", f32, "n"), T("target", f32, "n"), I("n"), F("weight"), T("grad_out", f32, 1, True),
                                     T("dx", f32, "n"), I("accumulate"), S],
    # ---- flat fp32 passes
    "udaseg_adam_flat": [T("p", f32, "count"), T("g", f32, "count"), T("m", f32, "count"), T("v", f32, "count"), I("count"), F("lr"),
                         F("beta1"), F("beta2"), F("eps"), F("bc1"), F("bc2"), S],
    "udaseg_fill_f32": [T("p", f32, "count"), I("count"), F("value"), S],
    "udaseg_axpy_f32": [T("y", f32, "count"), T("x", f32, "count"), I("count"), F("alpha"), S],
    "udaseg_add_i64": [T("p", i64, "count"), I("count"), I("value"), S],
    "udaseg_scale_f32": [T("x", f32, "count"), T("y", f32, "count"), I("count"), F("alpha"), S],
}

REQUIRE_CUDA = True          # tests/test_abi.py clears it to drive the checks with CPU tensors and a stubbed library
_FN = {}                     # entry point -> bound ctypes function (resolved on first use; the test puts stubs here)
_DESC_FIELDS = ("n", "hi", "wi", "ci", "ho", "wo", "co", "kh", "kw", "stride", "pad")
_CONST = {}


def _resolve(name):
    fn = _FN[name] = getattr(_lib.load(), name)
    return fn


def _R():
    r = _CONST.get("R")
    if r is None:
        r = _CONST["R"] = int(_lib.load().udaseg_bn_replicas())
    return r


def _memo(key, fn, *a):
    v = _CONST.get((key,) + a)
    if v is None:
        v = _CONST[(key,) + a] = int(fn(*a))
    return v


def _frag(n_out, k_in, ks):
    return _memo("frag", _lib.load().udaseg_frag_elems, n_out, k_in, ks)


def _gap_splits(hw):
    return _memo("gap", _lib.load().udaseg_gap_splits, hw)


def _ce_partials():
    return _memo("ce", _lib.load().udaseg_ce_partials)


def _seg_partials():
    return _memo("seg", _lib.load().udaseg_seg_partials)


def _current_stream():
    return torch.cuda.current_stream().cuda_stream


def _bad(entry, role, t, want, need, dev):
    """Says which of the operand's properties is wrong (called only on the failing path)."""
    if not torch.is_tensor(t):
        raise ValueError(f"{entry}: {role} must be a torch.Tensor{' (required)' if t is None else ''}, got {type(t).__name__}")
    if want == "raw32":
        have = f"{t.numel() * t.element_size() // 4} 4-byte units"
    else:
        have = f"{t.numel()} elements"
        if t.dtype is not want:
            raise ValueError(f"{entry}: {role} must be {want} (the kernel reads its bytes as that), got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{entry}: {role} must be contiguous")
    if REQUIRE_CUDA and not t.is_cuda:
        raise ValueError(f"{entry}: {role} must live on the GPU, got {t.device}")
    if dev is not None and t.device != dev:
        raise ValueError(f"{entry}: {role} is on {t.device}, the call's other tensors on {dev}")
    raise ValueError(f"{entry}: {role} is too short: the kernel touches {need} elements, the tensor has {have}")


def _compile(entry, roles):
    """Source of the checked caller of one entry point (readable with ops.source(name))."""
    names = []
    for r in roles:
        names.append({"desc": "d", "stream": "stream"}.get(r[0]) or r[1])
    src = [f"def {entry}({', '.join(names)}):"]
    exprs = " ".join(r[2] + " " + r[3] for r in roles if r[0] == "tensor" and isinstance(r[2], str)) + " " + \
            " ".join(r[3] for r in roles if r[0] == "tensor")
    import re
    used = set(re.findall(r"[A-Za-z_]\w*", exprs))
    if any(r is D for r in roles):
        need = set(used)
        if "X" in used:
            need |= {"n", "hi", "wi", "ci"}
        if "Y" in used:
            need |= {"n", "ho", "wo", "co"}
        if "W" in used:
            need |= {"co", "kh", "kw", "ci"}
        for f in _DESC_FIELDS:
            if f in need:
                src.append(f"    {f} = d.{f}")
        if "X" in used:
            src.append("    X = n * hi * wi * ci")
        if "Y" in used:
            src.append("    Y = n * ho * wo * co")
        if "W" in used:
            src.append("    W = co * kh * kw * ci")
    if "R" in used:
        src.append("    R = _R()")
    src.append("    dev = None")
    cargs = []
    for r in roles:
        kind = r[0]
        if kind == "desc":
            cargs.append("_byref(d)")
        elif kind == "stream":
            cargs.append("stream if stream is not None else _current_stream()")
        elif kind in ("int", "float", "host"):
            cargs.append(r[1])
        else:
            _, nm, dt, cnt, opt = r
            want = dt if isinstance(dt, str) and dt not in _DT else None
            dts = dt if want else {v: k for k, v in _DT.items()}.get(dt, dt)
            ind = "    "
            if opt:
                src.append(f"    if {nm} is not None:")
                ind = "        "
            src.append(f"{ind}_need = {cnt}")
            if dts == "raw32":
                src.append(f"{ind}if (not _is_tensor({nm}) or {nm}.numel() * {nm}.element_size() < 4 * _need or not {nm}.is_contiguous()")
                src.append(f"{ind}        or (REQUIRE_CUDA and not {nm}.is_cuda) or (dev is not None and {nm}.device != dev)):")
                src.append(f"{ind}    _bad('{entry}', '{nm}', {nm}, 'raw32', _need, dev)")
            else:
                src.append(f"{ind}_want = {dts}")
                src.append(f"{ind}if (not _is_tensor({nm}) or {nm}.dtype is not _want or {nm}.numel() < _need or not {nm}.is_contiguous()")
                src.append(f"{ind}        or (REQUIRE_CUDA and not {nm}.is_cuda) or (dev is not None and {nm}.device != dev)):")
                src.append(f"{ind}    _bad('{entry}', '{nm}', {nm}, _want, _need, dev)")
            src.append(f"{ind}if dev is None:")
            src.append(f"{ind}    dev = {nm}.device")
            if opt:
                cargs.append(f"None if {nm} is None else {nm}.data_ptr()")
            else:
                cargs.append(f"{nm}.data_ptr()")
    src.append(f"    fn = _FN.get('{entry}') or _resolve('{entry}')")
    src.append(f"    return fn({', '.join(cargs)})")
    return "\n".join(src) + "\n"


class _Ops:
    """ops.<entry point>(*args in C order; tensors as tensors) -> the library's return value, after the checks."""

    def __init__(self):
        self._src = {}
        g = {"_R": _R, "_bad": _bad, "_FN": _FN, "_resolve": _resolve, "_byref": C.byref, "_current_stream": _current_stream,
             "_is_tensor": torch.is_tensor, "frag": _frag, "gap_splits": _gap_splits, "ce_partials": _ce_partials,
             "seg_partials": _seg_partials, **_DT}
        self._globals = g
        for entry, roles in OPERANDS.items():
            src = _compile(entry, roles)
            self._src[entry] = src
            exec(compile(src, f"<operands:{entry}>", "exec"), g)
            setattr(self, entry, g[entry])

    def source(self, entry):
        return self._src[entry]


ops = _Ops()


def requirements(entry, *args):
    """[(role name, dtype or "raw32", fewest elements, optional)] of the tensor operands of one call -- the table's expressions
    evaluated directly (tests/test_abi.py builds its operands from this and holds the generated callers to it)."""
    roles = OPERANDS[entry]
    if len(args) != len(roles):
        raise TypeError(f"{entry}: {len(roles)} arguments expected, got {len(args)}")
    env = {"frag": _frag, "gap_splits": _gap_splits, "ce_partials": _ce_partials, "seg_partials": _seg_partials, **_DT}
    for r, a in zip(roles, args):
        if r[0] == "desc":
            for f in _DESC_FIELDS:
                env[f] = getattr(a, f)
            env["X"] = a.n * a.hi * a.wi * a.ci
            env["Y"] = a.n * a.ho * a.wo * a.co
            env["W"] = a.co * a.kh * a.kw * a.ci
        elif r[0] in ("int", "float"):
            env[r[1]] = a
    env["R"] = _R()
    out = []
    for r in roles:
        if r[0] == "tensor":
            dt = r[2]
            if isinstance(dt, str) and dt != "raw32":
                dt = eval(dt, {}, env)
            out.append((r[1], dt, int(eval(r[3], {}, env)), r[4]))
    return out


def set_require_cuda(flag):
    """tests only: accept CPU tensors (the device check is the last one a CPU tensor could pass otherwise)."""
    global REQUIRE_CUDA
    REQUIRE_CUDA = bool(flag)
    ops._globals["REQUIRE_CUDA"] = REQUIRE_CUDA


ops._globals["REQUIRE_CUDA"] = REQUIRE_CUDA
