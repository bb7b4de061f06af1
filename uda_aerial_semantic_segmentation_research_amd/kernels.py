"""Thin tensor-level wrappers over the C-ABI (one Python function per entry point of include/udaseg.h).

No autograd here and no arithmetic: these marshal ``torch.Tensor`` storage pointers, sizes and the current HIP
stream into libudaseg_hip.so.  All activations are NHWC fp32 contiguous tensors ``[N, H, W, C]`` with C % 4 == 0.
"""
import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ConvDesc, check
from ._operands import OPERANDS, ops          # the table every pointer-passing call goes through (checks before data_ptr())

_byref = _lib.C.byref


def stream():
    """hipStream_t of torch's current stream (kernels are enqueued there, asynchronously)."""
    return torch.cuda.current_stream().cuda_stream


def zeros(shape, dtype, device, st=None):
    """torch.zeros at a sixth of its host cost: torch.empty + one hipMemsetAsync on the plan's stream (a torch fill costs ~30 us of
    host time on this stack, and BASELINE cfg 3 is bound by the host's launch rate: profiles/r04_host_bound.txt)."""
    t = torch.empty(shape, dtype=dtype, device=device)
    check(ops.udaseg_memset_async(t, 0, t.numel() * t.element_size(), st), "memset_async")
    return t


def zeros_like(x, st=None):
    return zeros(x.shape, x.dtype, x.device, st)


def stream_wait(waiter, signal):
    """Stream ``waiter`` (a hipStream_t handle) waits for everything enqueued on ``signal`` so far -- torch's
    ``Event().record(signal); waiter.wait_event(ev)`` at a third of its host cost (a pooled event inside the library)."""
    check(ops.udaseg_stream_wait(waiter, signal), "stream_wait")


_WORKSPACE = {}


def ensure_workspace(device, nbytes=16 << 20):
    """Hand the library its scratch buffer (once per device; PyTorch owns the memory)."""
    key = str(device)
    if key not in _WORKSPACE:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        scr = torch.zeros(256 << 10, dtype=torch.uint8, device=device)     # zeroed once; the library keeps it zeroed
        with torch.cuda.device(device):          # the library binds the buffers to the CURRENT device
            check(ops.udaseg_set_workspace(buf, nbytes), "set_workspace")
            check(ops.udaseg_set_stats_scratch(scr, scr.numel()), "set_stats_scratch")
        _WORKSPACE[key] = buf
        _WORKSPACE[key + "/stats"] = scr
    return _WORKSPACE[key]


def conv_desc(n, hi, wi, ci, co, k, stride, pad):
    ho = (hi + 2 * pad - k) // stride + 1
    wo = (wi + 2 * pad - k) // stride + 1
    return ConvDesc(n, hi, wi, ci, ho, wo, co, k, k, stride, pad)


def conv2d_fwd(d, x, w, bias, y, act=ACT_NONE, slope=0.0, accumulate=False, st=None):
    if x.dtype == torch.bfloat16:
        assert not accumulate
        return conv2d_fwd_bf16(d, x, w, bias, None, y, act, slope, None, st)
    check(ops.udaseg_conv2d_fwd(d, x, w, bias, y, act, slope,
                                         int(accumulate), st), "conv2d_fwd")


def conv2d_fwd_bnstats(d, x, w, bias, y, stats, st=None):
    if x.dtype == torch.bfloat16:
        return conv2d_fwd_bf16(d, x, w, bias, None, y, ACT_NONE, 0.0, stats, st)
    check(ops.udaseg_conv2d_fwd_bnstats(d, x, w, bias, y,
                                                 stats, st), "conv2d_fwd_bnstats")


def conv2d_fwd_fused(d, x, w, bias, residual, y, act=ACT_NONE, slope=0.0, st=None):
    if x.dtype == torch.bfloat16:
        return conv2d_fwd_bf16(d, x, w, bias, residual, y, act, slope, None, st)
    check(ops.udaseg_conv2d_fwd_fused(d, x, w, bias, residual, y,
                                               act, slope, st), "conv2d_fwd_fused")


def bn_fold(w, bias, gamma, beta, running_mean, running_var, eps, w_folded, bias_folded, st=None):
    co = w.shape[0]
    check(ops.udaseg_bn_fold(w, bias, gamma, beta, running_mean,
                                      running_var, eps, co, w.numel() // co, w_folded,
                                      bias_folded, st), "bn_fold")


def argmax_confusion(logits_base, target, pixels, classes, ldc, confusion, pred=None, st=None):
    check(ops.udaseg_argmax_confusion(logits_base, target, pixels, classes, ldc,
                                               confusion, pred, st),
          "argmax_confusion")


def conv2d_fwd_bf16(d, x, w, bias, residual, y, act=ACT_NONE, slope=0.0, stats=None, st=None):
    """bf16 x / w / residual; y bf16, or fp32 when its dtype says so (logits)."""
    check(ops.udaseg_conv2d_fwd_bf16(d, x, w, bias, residual, y,
                                              int(y.dtype == torch.float32), act, slope, stats,
                                              st), "conv2d_fwd_bf16")


def conv2d_dgrad_bf16(d, dy, w_t, dx, accumulate=False, st=None):
    check(ops.udaseg_conv2d_dgrad_bf16(d, dy, w_t, dx, int(accumulate),
                                                st), "conv2d_dgrad_bf16")


def conv2d_wgrad_bf16(d, x, dy, dw, accumulate=False, st=None):
    check(ops.udaseg_conv2d_wgrad_bf16(d, x, dy, dw, int(accumulate),
                                                st), "conv2d_wgrad_bf16")


def conv2d_dgrad(d, dy, w_t, dx, accumulate=False, st=None):
    if dy.dtype == torch.bfloat16:
        return conv2d_dgrad_bf16(d, dy, w_t, dx, accumulate, st)
    check(ops.udaseg_conv2d_dgrad(d, dy, w_t, dx, int(accumulate),
                                           st), "conv2d_dgrad")


def conv2d_dgrad_bnreduce_ok(d, dtype=torch.float32):
    if dtype == torch.bfloat16:
        return bool(ops.udaseg_conv2d_dgrad_bnreduce_bf16_ok(d))
    return dtype == torch.float32 and bool(ops.udaseg_conv2d_dgrad_bnreduce_ok(d))


def conv2d_dgrad_bnreduce(d, dy, w_t, dx, prev_y, save_mean, save_rstd, gamma, beta, act, slope, bsums, st=None):
    """dx = dgrad AND the BatchNorm-backward reductions (sum g, sum g*xhat) of the layer whose output prev_y feeds this conv."""
    fn = ops.udaseg_conv2d_dgrad_bnreduce_bf16 if dy.dtype == torch.bfloat16 else ops.udaseg_conv2d_dgrad_bnreduce
    check(fn(d, dy, w_t, dx, prev_y, save_mean,
             save_rstd, gamma, beta, act, slope, bsums,
             st), "conv2d_dgrad_bnreduce")


def conv2d_wgrad(d, x, dy, dw, accumulate=False, st=None):
    if x.dtype == torch.bfloat16:
        return conv2d_wgrad_bf16(d, x, dy, dw, accumulate, st)
    check(ops.udaseg_conv2d_wgrad(d, x, dy, dw, int(accumulate),
                                           st), "conv2d_wgrad")


def conv2d_fwd_upcat(d, a, skip, w, bias, y, act=ACT_NONE, slope=0.0, stats=None, st=None):
    """y = act(conv(cat([nearest_x2(a), skip], C), w) + bias) without materialising the concatenation (+ BN statistics of y)."""
    fn = ops.udaseg_conv2d_fwd_upcat_bf16 if a.dtype == torch.bfloat16 else ops.udaseg_conv2d_fwd_upcat
    check(fn(d, a, skip, a.shape[-1], w, bias, y, act, slope, stats,
             st), "conv2d_fwd_upcat")


def conv2d_dgrad_split(d, dy, w_t, dx_a, dx_b, st=None):
    fn = ops.udaseg_conv2d_dgrad_split_bf16 if dy.dtype == torch.bfloat16 else ops.udaseg_conv2d_dgrad_split
    check(fn(d, dy, w_t, dx_a, dx_b, dx_a.shape[-1],
             st), "conv2d_dgrad_split")


def conv2d_wgrad_part(d, src, c_off, up, dy, dw, accumulate=True, st=None):
    fn = ops.udaseg_conv2d_wgrad_part_bf16 if src.dtype == torch.bfloat16 else ops.udaseg_conv2d_wgrad_part
    check(fn(d, src, src.shape[-1], c_off, int(up), dy, dw, int(accumulate),
             st), "conv2d_wgrad_part")


def upcat_fusable(ca, cb, co, dtype):
    """Can the convolution over cat([up(a), skip]) take the fused gather?  Channel counts that are multiples of the K-tile
    for the implicit-GEMM kernels (32 fp32 / 64 bf16) plus a 64-channel boundary for the data gradient's two outputs, or the
    fp32 small-channel kernel (<= 32 channels in all, no skip)."""
    if dtype == torch.bfloat16:
        return ca % 64 == 0 and cb % 64 == 0
    g_in, g_out = (ca + cb + 15) // 16, (co + 15) // 16
    if cb == 0 and g_in * g_out <= 2:
        return True
    return ca % 32 == 0 and cb % 32 == 0 and (cb == 0 or ca % 64 == 0)


def bn_finalize(sums, gamma, beta, pixels, eps, momentum, running_mean, running_var, save_mean, save_rstd, scale, shift, st=None):
    """Statistics -> saved mean / rstd, running statistics and the per-channel scale / shift a consumer applies while staging."""
    c = scale.numel()
    check(ops.udaseg_bn_finalize(sums, gamma, beta, pixels, c, eps, momentum,
                                          running_mean, running_var, save_mean, save_rstd, scale,
                                          shift, st), "bn_finalize")


def bn_bwd_apply_recompute(dz, y, fwd_scale, fwd_shift, save_mean, save_rstd, gamma, bsums, dy, dgamma, dbeta, act, slope, st=None):
    """BatchNorm backward of a layer whose activation was never written: the mask is re-evaluated from y, fwd_scale, fwd_shift."""
    c = y.shape[-1]
    if y.dtype != torch.bfloat16:
        raise ValueError("bn_bwd_apply_recompute: bf16 storage only")
    check(ops.udaseg_bn_bwd_apply_recompute_bf16(dz, y, fwd_scale, fwd_shift,
                                                          save_mean, save_rstd, gamma,
                                                          bsums, dy, dgamma, dbeta, y.numel() // c,
                                                          c, act, slope, st),
          "bn_bwd_apply_recompute_bf16")


def conv2d_wgrad_halo_ok(d, up_ca=0, f32=False):
    if f32:
        return bool(ops.udaseg_conv2d_wgrad_halo_f32x3_ok(d, up_ca))
    return bool(ops.udaseg_conv2d_wgrad_halo_bf16_ok(d, up_ca))


def conv2d_wgrad_halo(d, x, skip, dy, dw, up=False, st=None):
    """dW += weight gradient of a stride-1 3x3 layer (channel counts multiples of 64) on the halo-resident kernel: bf16 tensors,
    or fp32 tensors with the exact three-term split.  up: x is the half-resolution source of a fused decoder input, skip the
    other one."""
    fn, name = ((ops.udaseg_conv2d_wgrad_halo_f32x3, "conv2d_wgrad_halo_f32x3") if x.dtype == torch.float32
                else (ops.udaseg_conv2d_wgrad_halo_bf16, "conv2d_wgrad_halo_bf16"))
    check(fn(d, x, skip, x.shape[-1] if up else 0, dy, dw,
             st), name)


def conv_bnin_ok(d, up=False):
    """fp32: can BOTH consumers of an unwritten BatchNorm activation in front of this convolution apply it while staging?
    (forward on the split kernels, weight gradient on the small-channel direct kernel or the halo-resident split kernel)
    up: the activation reaches the convolution through a nearest x2 up-sampling (d describes the up-sampled geometry)."""
    return bool(ops.udaseg_conv2d_fwd_f32x3_bnin_ok(d, int(up))) and bool(ops.udaseg_conv2d_wgrad_bnin_ok(d, int(up)))


def conv_bnin_writes(d):
    """fp32: the forward of this convolution can write the transformed activation out while it stages it (wave-specialised kernel)"""
    return bool(ops.udaseg_conv2d_fwd_f32x3_bnin_writes(d)) and bool(ops.udaseg_conv2d_fwd_f32x3_bnin_ok(d, 0))


def conv2d_wgrad_bnin(d, y_prev, in_scale, in_shift, in_act, in_slope, dy, dw, accumulate=False, st=None, up=False):
    """Weight gradient whose gathered operand is act(fma(y_prev, in_scale, in_shift)) (never written), bf16 or fp32."""
    if y_prev.dtype == torch.float32:
        check(ops.udaseg_conv2d_wgrad_bnin(d, y_prev, int(up), in_scale, in_shift, in_act, in_slope, dy, dw, int(accumulate),
                                           st), "conv2d_wgrad_bnin")
        return
    check(ops.udaseg_conv2d_wgrad_bnin_bf16(d, y_prev, in_scale, in_shift, in_act,
                                                     in_slope, dy, dw, int(accumulate),
                                                     st), "conv2d_wgrad_bnin_bf16")


def frag_elems(n_out, k_in, ks):
    """bf16 elements of the MFMA-fragment packing of a convolution with n_out produced / k_in gathered channels, ks x ks window."""
    return int(ops.udaseg_frag_elems(n_out, k_in, ks))


def pack_frag_batched(w, wt, packed, table, st=None):
    """Fragment-pack every listed convolution in one launch (table rows: mode, src offset, dst offset, N, K, ks; int32 x 6).
    bf16 sources: one plane (csrc/conv_halo_bf16.hip); fp32 sources: the three split planes (csrc/conv_halo_f32x3.hip)."""
    src = w if w is not None else wt
    if src.dtype == torch.float32:
        check(ops.udaseg_pack_frag_batched_f32x3(w, wt, packed, table, table.shape[0],
                                                          st), "pack_frag_batched_f32x3")
        return
    check(ops.udaseg_pack_frag_batched_bf16(w, wt, packed, table, table.shape[0],
                                                     st), "pack_frag_batched_bf16")


def pack_up_batched(w, wt, packed, table, st=None):
    """Fragment packings of the decoder conv1 layers whose up-sampled half runs as four 2x2 phase convolutions
    (csrc/conv_up_f32x3.hip): table rows of 8 int32 {mode, src offset, dst offset, N, K, ldk, 0, 0}."""
    check(ops.udaseg_pack_up_batched_f32x3(w, wt, packed, table, table.shape[0], st), "pack_up_batched_f32x3")


def conv_up_ok(d, up_ca):
    return bool(ops.udaseg_conv_up_f32x3_ok(d, up_ca))


def conv2d_fwd_up(d, a, wfrag_up, y, accumulate=False, stats=None, st=None):
    """y (+)= conv3x3(nearest_x2(a)) as four 2x2 phase convolutions of a (fp32, three-term split); d: the whole decoder conv1."""
    check(ops.udaseg_conv2d_fwd_up_f32x3(d, a, a.shape[-1], wfrag_up, y, int(accumulate), stats, st), "conv2d_fwd_up_f32x3")


def conv2d_dgrad_up(d, dy, up_ca, wfrag_up_t, da, accumulate=False, bn=None, st=None):
    """da (+)= gradient of a through conv3x3(nearest_x2(a)), at a's resolution.
    bn = (prev_y, save_mean, save_rstd, gamma, beta, act, slope, bsums): BatchNorm-backward reductions of the layer that produced a."""
    py, mu, rs, ga, be, act, slope, bs = bn if bn is not None else (None, None, None, None, None, ACT_NONE, 0.0, None)
    check(ops.udaseg_conv2d_dgrad_up_f32x3(d, dy, up_ca, wfrag_up_t, da, py, mu, rs, ga, be, act, slope, bs, int(accumulate), st),
          "conv2d_dgrad_up_f32x3")


def conv2d_wgrad_up_ok(d, up_ca):
    return bool(ops.udaseg_conv2d_wgrad_up_f32x3_ok(d, up_ca))


def conv2d_wgrad_up(d, a, dy, dw, st=None):
    """dW[..., :ca] += weight gradient of conv3x3(nearest_x2(a)) in the phase form (16 phase taps at a's resolution); d: the whole
    decoder conv1, dW its [co][3][3][ci] gradient."""
    check(ops.udaseg_conv2d_wgrad_up_f32x3(d, a, a.shape[-1], dy, dw, st), "conv2d_wgrad_up_f32x3")


def conv2d_wgrad_halo_slice(d_slice, x, dy, dw, c_off, st=None):
    """dW[..., c_off:c_off + x channels] += halo-resident weight gradient of the slice (d_slice: the slice as its own convolution)."""
    check(ops.udaseg_conv2d_wgrad_halo_slice_f32x3(d_slice, x, dy, dw, dw.shape[-1], c_off, st), "conv2d_wgrad_halo_slice_f32x3")


STEM_FRAG_ELEMS = 3 * 28 * 512


def conv_stem_ok(d):
    return bool(ops.udaseg_conv_stem_f32x3_ok(d))


def conv2d_fwd_stem(d, x, wfrag, y, stats=None, st=None):
    """The encoder's 7x7 / stride 2 stem on its own kernel (csrc/conv_stem_f32x3.hip, mode-8 packing)."""
    check(ops.udaseg_conv2d_fwd_stem_f32x3(d, x, wfrag, y, stats, st), "conv2d_fwd_stem_f32x3")


def conv_n16_ok(d, dgrad=False):
    return bool(ops.udaseg_conv_n16_f32x3_ok(d, int(dgrad)))


def n16_frag_elems(k_in):
    """bf16 elements of the three-plane sixteen-wide-tile packing of a convolution gathering k_in channels."""
    return 3 * ((k_in + 15) // 16) * 5 * 512


def conv2d_fwd_n16(d, x, wfrag, y, stats=None, in_scale=None, in_shift=None, in_act=ACT_NONE, in_slope=0.0, st=None):
    """Forward 3x3 convolution producing 16 channels on the sixteen-wide matrix tile (csrc/conv_n16_f32x3.hip)."""
    check(ops.udaseg_conv2d_fwd_n16_f32x3(d, x, in_scale, in_shift, in_act, in_slope, wfrag, y, stats, st), "conv2d_fwd_n16_f32x3")


def conv2d_dgrad_n16(d, dy, wfrag_t, dx, bn=None, st=None):
    """Data gradient of a 3x3 convolution with 16 input channels on the sixteen-wide tile; bn as conv2d_dgrad_frag."""
    py, mu, rs, ga, be, act, slope, bs = bn if bn is not None else (None, None, None, None, None, ACT_NONE, 0.0, None)
    check(ops.udaseg_conv2d_dgrad_n16_f32x3(d, dy, wfrag_t, dx, py, mu, rs, ga, be, act, slope, bs, st), "conv2d_dgrad_n16_f32x3")


def conv_frag_ok(d, dgrad=False, up_ca=0, f32=False):
    if f32:
        return bool(ops.udaseg_conv_f32x3_ok(d, int(dgrad), up_ca))
    return bool(ops.udaseg_conv_frag_ok(d, int(dgrad), up_ca))


def conv_frag_preferred(d, dgrad=False, up_ca=0, f32=False):
    """Supported AND expected to beat the shared implicit-GEMM kernel for this shape (the library's measured heuristic)."""
    if f32:
        return bool(ops.udaseg_conv_f32x3_preferred(d, int(dgrad), up_ca))
    return bool(ops.udaseg_conv_frag_preferred(d, int(dgrad), up_ca))


def conv2d_fwd_frag(d, x, skip, wfrag, bias, y, act=ACT_NONE, slope=0.0, stats=None, in_scale=None, in_shift=None, in_act=ACT_NONE,
                    in_slope=0.0, up=False, st=None, z_out=None):
    """Halo-resident forward convolution on the bf16 matrix pipe: bf16 tensors (csrc/conv_halo_bf16.hip) or fp32 tensors with
    the three-term split (csrc/conv_halo_f32x3.hip).  up: x is the half-resolution tensor of a fused decoder input."""
    if x.dtype == torch.float32 and in_scale is not None:
        assert y.dtype == torch.float32 and skip is None
        check(ops.udaseg_conv2d_fwd_f32x3_bnin(d, x, int(up), in_scale, in_shift, in_act, in_slope, z_out, wfrag, bias, y, act, slope,
                                               stats, st), "conv2d_fwd_f32x3_bnin")
        return
    assert z_out is None
    if x.dtype == torch.float32:
        assert y.dtype == torch.float32
        check(ops.udaseg_conv2d_fwd_f32x3(d, x, skip, x.shape[-1] if up else 0, wfrag,
                                                   bias, y, act, slope, stats,
                                                   st), "conv2d_fwd_f32x3")
        return
    check(ops.udaseg_conv2d_fwd_frag_bf16(d, x, skip, x.shape[-1] if up else 0, wfrag,
                                                   bias, in_scale, in_shift, in_act, in_slope, y,
                                                   int(y.dtype == torch.float32), act, slope, stats,
                                                   st), "conv2d_fwd_frag_bf16")


def conv2d_dgrad_frag(d, dy, wfrag_t, dx, dx2=None, bn=None, accumulate=False, st=None):
    """Halo-resident data gradient (bf16 tensors, or fp32 tensors with the three-term split).  dx2: second destination of a
    split gradient (channels [dx.shape[-1], ci)).
    bn = (prev_y, save_mean, save_rstd, gamma, beta, act, slope, bsums): BatchNorm-backward reductions of the layer behind."""
    py, mu, rs, ga, be, act, slope, bs = bn if bn is not None else (None, None, None, None, None, ACT_NONE, 0.0, None)
    fn, name = ((ops.udaseg_conv2d_dgrad_f32x3, "conv2d_dgrad_f32x3") if dy.dtype == torch.float32
                else (ops.udaseg_conv2d_dgrad_frag_bf16, "conv2d_dgrad_frag_bf16"))
    check(fn(d, dy, wfrag_t, dx, dx2, dx.shape[-1] if dx2 is not None else 0,
             py, mu, rs, ga, be, act, slope, bs, int(accumulate),
             st), name)


def pack_dgrad_weights(d, w, w_t, st=None):
    check(ops.udaseg_pack_dgrad_weights(d, w, w_t,
                                                 st), "pack_dgrad_weights")


def cast_to_bf16(x, out=None, st=None):
    """fp32 -> bf16 (round to nearest even), flat; numel must be a multiple of 8."""
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(ops.udaseg_cast_f32_to_bf16(x, out, x.numel(), st),
          "cast_f32_to_bf16")
    return out


def pack_dgrad_batched(arena, packed, table, st=None):
    if packed.dtype == torch.bfloat16:
        check(ops.udaseg_pack_dgrad_batched_bf16(arena, packed, table, table.shape[0],
                                                          st), "pack_dgrad_batched_bf16")
        return
    check(ops.udaseg_pack_dgrad_batched(arena, packed, table, table.shape[0],
                                                 st), "pack_dgrad_batched")


def conv_flops(d):
    return ops.udaseg_conv_flops(d)


def nchw_to_nhwc(x, cpad=None, st=None, dtype=torch.float32):
    """[N,C,H,W] contiguous fp32 -> [N,H,W,cpad] (zero-padded channels), fp32 or bf16."""
    n, c, h, w = x.shape
    if not x.is_contiguous():
        x = x.contiguous()
    if dtype == torch.bfloat16:
        cpad = cpad or ((c + 7) // 8) * 8
        y = torch.empty((n, h, w, cpad), device=x.device, dtype=torch.bfloat16)
        check(ops.udaseg_nchw_to_nhwc_bf16(x, y, n, c, h, w, cpad,
                                                    st), "nchw_to_nhwc_bf16")
        return y
    cpad = cpad or ((c + 3) // 4) * 4
    y = torch.empty((n, h, w, cpad), device=x.device, dtype=torch.float32)
    check(ops.udaseg_nchw_to_nhwc(x, y, n, c, h, w, cpad,
                                           st), "nchw_to_nhwc")
    return y


_BN_R = None


def bn_replicas():
    global _BN_R
    if _BN_R is None:
        _BN_R = ops.udaseg_bn_replicas()
    return _BN_R


def bn_stats(y, sums, st=None):
    c = y.shape[-1]
    if y.dtype not in (torch.float32, torch.bfloat16) or not y.is_contiguous():
        raise ValueError(f"bn_stats: contiguous fp32 or bf16 tensor expected, got {y.dtype}")
    fn = ops.udaseg_bn_stats_bf16 if y.dtype == torch.bfloat16 else ops.udaseg_bn_stats
    check(fn(y, y.numel() // c, c, sums, st), "bn_stats")


def bn_apply(y, sums, gamma, beta, residual, z, eps, momentum, running_mean, running_var, save_mean, save_rstd, act, slope,
             st=None):
    c = y.shape[-1]
    if y.dtype == torch.bfloat16:
        check(ops.udaseg_bn_apply_bf16(y, sums, gamma, beta, residual,
                                                z, y.numel() // c, c, eps, momentum, running_mean,
                                                running_var, save_mean, save_rstd, act, slope,
                                                st), "bn_apply_bf16")
        return
    check(ops.udaseg_bn_apply(y, sums, gamma, beta, residual,
                                       z, y.numel() // c, c, eps, momentum, running_mean, running_var,
                                       save_mean, save_rstd, act, slope,
                                       st), "bn_apply")


def bn_apply_eval(y, gamma, beta, running_mean, running_var, residual, z, eps, act, slope, st=None):
    c = y.shape[-1]
    check(ops.udaseg_bn_apply_eval(y, gamma, beta, running_mean,
                                            running_var, residual, z, y.numel() // c, c, eps, act,
                                            slope, st), "bn_apply_eval")


def bn_bwd_reduce(dz, z, y, save_mean, save_rstd, bsums, act, slope, st=None, gamma=None, beta=None):
    """z=None (fp32 only, layers without a residual input): the activation's argument is re-evaluated from y, gamma, beta."""
    c = y.shape[-1]
    if y.dtype == torch.bfloat16:
        check(ops.udaseg_bn_bwd_reduce_bf16(dz, z, y, save_mean,
                                                     save_rstd, y.numel() // c, c, bsums, act, slope,
                                                     st), "bn_bwd_reduce_bf16")
        return
    check(ops.udaseg_bn_bwd_reduce(dz, z, y, save_mean, save_rstd,
                                            gamma, beta, y.numel() // c, c, bsums, act, slope,
                                            st), "bn_bwd_reduce")


def bn_bwd_apply(dz, z, y, save_mean, save_rstd, gamma, bsums, dy, dres, dgamma, dbeta, act, slope, accumulate_dy=False,
                 accumulate_dres=False, accumulate_param=False, st=None, beta=None):
    c = y.shape[-1]
    if y.dtype == torch.bfloat16:
        check(ops.udaseg_bn_bwd_apply_bf16(dz, z, y, save_mean,
                                                    save_rstd, gamma, bsums, dy,
                                                    dres, dgamma, dbeta, y.numel() // c, c, act, slope,
                                                    int(accumulate_dy), int(accumulate_dres), int(accumulate_param),
                                                    st), "bn_bwd_apply_bf16")
        return
    check(ops.udaseg_bn_bwd_apply(dz, z, y, save_mean, save_rstd,
                                           gamma, beta, bsums, dy, dres, dgamma,
                                           dbeta, y.numel() // c, c, act, slope, int(accumulate_dy),
                                           int(accumulate_dres), int(accumulate_param),
                                           st), "bn_bwd_apply")


def act_bwd(dz, z, dy, act, slope, st=None):
    if dz.dtype == torch.bfloat16:
        check(ops.udaseg_act_bwd_bf16(dz, z, dy, dz.numel(), act, slope,
                                               st), "act_bwd_bf16")
        return
    check(ops.udaseg_act_bwd(dz, z, dy, dz.numel(), act, slope,
                                      st), "act_bwd")


_CHSUM_SCRATCH = {}          # (device index, stream handle) -> persistent fp32 scratch for channel_sum's partial sums


def channel_sum(x, out, accumulate=False, st=None):
    """out[c] (+)= sum over pixels.  Large inputs reduce through 16 replicas of ``out`` in a scratch that belongs to the
    stream the call runs on (kept per stream: calls on one stream are ordered, calls on different streams never share it)."""
    c = x.shape[-1]
    st = st
    key = (x.device.index, st)
    ws = _CHSUM_SCRATCH.get(key)
    need = 16 * c
    if ws is None or ws.numel() < need:
        ws = _CHSUM_SCRATCH[key] = torch.empty(max(need, 16 * 2048), dtype=torch.float32, device=x.device)
    fn = ops.udaseg_channel_sum_bf16_ws if x.dtype == torch.bfloat16 else ops.udaseg_channel_sum_ws
    check(fn(x, x.numel() // c, c, out, int(accumulate), ws, ws.numel() * 4, st), "channel_sum")


def maxpool_fwd(x, st=None):
    n, h, w, c = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((n, ho, wo, c), device=x.device, dtype=x.dtype)
    idx = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8)
    if x.dtype == torch.bfloat16:
        check(ops.udaseg_maxpool3x3s2_fwd_bf16(x, y, idx, n, h, w, c,
                                                        st), "maxpool_fwd_bf16")
        return y, idx
    check(ops.udaseg_maxpool3x3s2_fwd(x, y, idx, n, h, w, c,
                                               st), "maxpool_fwd")
    return y, idx


def maxpool_bwd(dy, idx, dx, accumulate=False, st=None):
    n, h, w, c = dx.shape
    if dx.dtype == torch.bfloat16:
        check(ops.udaseg_maxpool3x3s2_bwd_bf16(dy, idx, dx, n, h, w, c, int(accumulate),
                                                        st), "maxpool_bwd_bf16")
        return
    check(ops.udaseg_maxpool3x3s2_bwd(dy, idx, dx, n, h, w, c, int(accumulate),
                                               st), "maxpool_bwd")


def upsample2x_concat_fwd(a, skip, st=None):
    n, h, w, ca = a.shape
    cb = 0 if skip is None else skip.shape[-1]
    out = torch.empty((n, 2 * h, 2 * w, ca + cb), device=a.device, dtype=a.dtype)
    if a.dtype == torch.bfloat16:      # pure data movement in 16-byte vectors: 8 bf16 channels == 4 fp32 "channels"
        ca, cb = ca // 2, cb // 2
    check(ops.udaseg_upsample2x_concat_fwd(a, skip, out, n, h, w, ca, cb,
                                                    st), "upsample2x_concat_fwd")
    return out


def upsample2x_concat_bwd(dout, da, dskip, ca, cb, accumulate_da=False, accumulate_dskip=False, st=None):
    n, h2, w2, _ = dout.shape
    if dout.dtype == torch.bfloat16:
        check(ops.udaseg_upsample2x_concat_bwd_bf16(dout, da, dskip, n, h2 // 2, w2 // 2, ca, cb,
                                                             int(accumulate_da), int(accumulate_dskip),
                                                             st), "upsample2x_concat_bwd_bf16")
        return
    check(ops.udaseg_upsample2x_concat_bwd(dout, da, dskip, n, h2 // 2, w2 // 2, ca, cb,
                                                    int(accumulate_da), int(accumulate_dskip),
                                                    st), "upsample2x_concat_bwd")


def upsample2x_bilinear_concat_fwd(a, skip, st=None):
    """cat(interpolate(a, scale_factor=2, mode='bilinear', align_corners=False), skip) on NHWC tensors (fp32 or bf16)."""
    n, h, w, ca = a.shape
    cb = 0 if skip is None else skip.shape[-1]
    out = torch.empty((n, 2 * h, 2 * w, ca + cb), device=a.device, dtype=a.dtype)
    check(ops.udaseg_upsample2x_bilinear_concat_fwd(a, skip, out, n, h, w, ca, cb,
                                                             int(a.dtype == torch.bfloat16), st),
          "upsample2x_bilinear_concat_fwd")
    return out


def upsample2x_bilinear_concat_bwd(dout, da, dskip, ca, cb, accumulate_da=False, accumulate_dskip=False, st=None):
    n, h2, w2, _ = dout.shape
    check(ops.udaseg_upsample2x_bilinear_concat_bwd(dout, da, dskip, n, h2 // 2, w2 // 2, ca, cb,
                                                             int(accumulate_da), int(accumulate_dskip),
                                                             int(dout.dtype == torch.bfloat16), st),
          "upsample2x_bilinear_concat_bwd")


def ce_fwd(logits_base, target, pixels, classes, ldc, lse, partials, loss, st=None):
    check(ops.udaseg_ce_fwd(logits_base, target, pixels, classes, ldc, lse,
                                     partials, loss, st), "ce_fwd")


def ce_bwd(logits_base, target, lse, grad_out, pixels, classes, ldc, dlogits, colsum_partials=None, colsum=None, st=None):
    check(ops.udaseg_ce_bwd(logits_base, target, lse, grad_out, pixels, classes,
                                     ldc, dlogits, colsum_partials, colsum,
                                     st), "ce_bwd")


def ce_fwd_bwd(logits_base, target, pixels, classes, ldc, partials, loss, dlogits, colsum_partials=None, colsum=None, st=None):
    """Loss and its gradient for an upstream gradient of 1 in ONE pass over the logits (ldc <= 32); both bit-identical to
    ``ce_fwd`` / ``ce_bwd``."""
    check(ops.udaseg_ce_fwd_bwd(logits_base, target, pixels, classes, ldc, partials, loss, dlogits, colsum_partials, colsum, st),
          "ce_fwd_bwd")


def scale_unless_one(x, g, x2=None, st=None):
    """x *= g (and x2 *= g) unless the device scalar ``g`` is exactly 1: then the launch returns at once."""
    check(ops.udaseg_scale_unless_one(x, x.numel(), x2, 0 if x2 is None else x2.numel(), g, st), "scale_unless_one")


def seg_partials():
    return ops.udaseg_seg_partials()


def dice_fwd(logits_base, target, batch, pix_per_image, classes, ldc, smooth, sums, coef, loss, eps=1e-7, pooled=False, st=None):
    check(ops.udaseg_dice_fwd(logits_base, target, batch, pix_per_image, classes, ldc,
                                       float(smooth), float(eps), int(pooled), sums, coef,
                                       loss, st), "dice_fwd")


def dice_bwd(logits_base, target, coef, grad_out, weight, batch, pix_per_image, classes, ldc, dlogits, accumulate=False,
             st=None):
    check(ops.udaseg_dice_bwd(logits_base, target, coef, grad_out,
                                       float(weight), batch, pix_per_image, classes, ldc, dlogits,
                                       int(accumulate), st), "dice_bwd")


def focal_fwd(logits_base, target, class_weights, alpha, gamma, pixels, classes, ldc, mean, partials, loss, accumulate=False,
              st=None):
    check(ops.udaseg_focal_fwd(logits_base, target, class_weights, float(alpha),
                                        float(gamma), pixels, classes, ldc, int(mean), partials, loss,
                                        int(accumulate), st), "focal_fwd")


def focal_bwd(logits_base, target, class_weights, alpha, gamma, grad_out, weight, pixels, classes, ldc, dlogits,
              accumulate=False, st=None):
    check(ops.udaseg_focal_bwd(logits_base, target, class_weights, float(alpha),
                                        float(gamma), grad_out, float(weight), pixels, classes, ldc,
                                        dlogits, int(accumulate), st), "focal_bwd")


def consistency_fwd(z1, z2, temperature, batch, pixels, classes, ldc, partials, loss, st=None):
    check(ops.udaseg_consistency_fwd(z1, z2, float(temperature), batch, pixels, classes, ldc,
                                              partials, loss, st),
          "consistency_fwd")


def consistency_bwd(z1, z2, temperature, grad_out, weight, batch, pixels, classes, ldc, d1, d2, accumulate=False, st=None):
    check(ops.udaseg_consistency_bwd(z1, z2, float(temperature), grad_out, float(weight),
                                              batch, pixels, classes, ldc, d1, d2, int(accumulate),
                                              st), "consistency_bwd")


def gap_linear_sigmoid_fwd(z, w, b, st=None):
    n, h, wd, c = z.shape
    hw = h * wd
    splits = ops.udaseg_gap_splits(hw)
    partial = torch.empty((n, splits, c), device=z.device, dtype=torch.float32)
    pooled = torch.empty((n, c), device=z.device, dtype=torch.float32)
    p = torch.empty((n, 1), device=z.device, dtype=torch.float32)
    if z.dtype == torch.bfloat16:
        sv = st
        check(ops.udaseg_gap_partial_bf16(z, partial, n, hw, c, sv), "gap_partial_bf16")
        check(ops.udaseg_gap_finish(partial, w, b, pooled, p, n, hw,
                                             c, sv), "gap_finish")
        return p, pooled
    check(ops.udaseg_gap_linear_sigmoid_fwd(z, w, b, partial,
                                                     pooled, p, n, hw, c,
                                                     st), "gap_linear_sigmoid_fwd")
    return p, pooled


def gap_linear_sigmoid_bwd(dp, p, pooled, w, dz, dw, db, accumulate_param=False, st=None):
    n, h, wd, c = dz.shape
    if dz.dtype == torch.bfloat16:
        sv = st
        check(ops.udaseg_gap_bwd_broadcast_bf16(dp, p, w, dz, n, h * wd, c, sv),
              "gap_bwd_broadcast_bf16")
        check(ops.udaseg_gap_bwd_param(dp, p, pooled, dw, db, n, c,
                                                int(accumulate_param), sv), "gap_bwd_param")
        return
    check(ops.udaseg_gap_linear_sigmoid_bwd(dp, p, pooled, w, dz,
                                                     dw, db, n, h * wd, c, int(accumulate_param),
                                                     st), "gap_linear_sigmoid_bwd")


def bce_logits_fwd(x, label, weight, loss, accumulate=False, st=None):
    check(ops.udaseg_bce_logits_fwd(x, x.numel(), label, weight, loss, int(accumulate),
                                             st), "bce_logits_fwd")


def bce_logits_bwd(x, label, weight, grad_out, dx, accumulate=False, st=None):
    check(ops.udaseg_bce_logits_bwd(x, x.numel(), label, weight, grad_out, dx,
                                             int(accumulate), st), "bce_logits_bwd")


def gap_linear_fwd(z, w, b, st=None):
    """logit[n] = dot(mean over pixels of z[n], w) + b for NHWC z; returns (logit [n], pooled [n, c])."""
    n, h, wd, c = z.shape
    hw = h * wd
    splits = ops.udaseg_gap_splits(hw)
    partial = torch.empty((n, splits, c), device=z.device, dtype=torch.float32)
    pooled = torch.empty((n, c), device=z.device, dtype=torch.float32)
    logit = torch.empty(n, device=z.device, dtype=torch.float32)
    check(ops.udaseg_gap_linear_fwd(z, w, b, partial, pooled,
                                             logit, n, hw, c, st), "gap_linear_fwd")
    return logit, pooled


def gap_linear_bwd(dlogit, pooled, w, dz, dw, db, accumulate_param=False, st=None):
    n, h, wd, c = dz.shape
    check(ops.udaseg_gap_linear_bwd(dlogit, pooled, w, dz, dw,
                                             db, n, h * wd, c, int(accumulate_param),
                                             st), "gap_linear_bwd")


def bce_logits_target_fwd(x, target, weight, loss, accumulate=False, st=None):
    check(ops.udaseg_bce_logits_target_fwd(x, target, x.numel(), float(weight), loss,
                                                    int(accumulate), st), "bce_logits_target_fwd")


def bce_logits_target_bwd(x, target, weight, grad_out, dx, accumulate=False, st=None):
    check(ops.udaseg_bce_logits_target_bwd(x, target, x.numel(), float(weight), grad_out,
                                                    dx, int(accumulate), st),
          "bce_logits_target_bwd")


def _storage_order(t, who):
    """A flat view of ``t`` in the order of its storage: ``t`` itself when contiguous, the underlying run of elements when ``t`` is a
    dense permutation (an [N,C,H,W]-shaped view of an NHWC buffer) -- an element-wise kernel does not care about the order."""
    if t.is_contiguous():
        return t
    st = sorted(zip(t.stride(), t.shape), reverse=True)
    run = 1
    for stride, size in reversed(st):
        if size != 1 and stride != run:
            raise ValueError(f"{who}: tensor must be dense (contiguous or a permutation of a contiguous tensor), strides {t.stride()}")
        run *= size
    return t.as_strided((t.numel(),), (1,), t.storage_offset())


def scale(x, alpha, out=None, st=None):
    """out = alpha * x (fp32; x dense in any dimension order, out with the same strides)."""
    out = torch.empty_like(x) if out is None else out
    if out.stride() != x.stride() or out.shape != x.shape:
        raise ValueError("scale: out must have x's shape and strides")
    check(ops.udaseg_scale_f32(_storage_order(x, "scale"), _storage_order(out, "scale"), x.numel(), float(alpha), st), "scale_f32")
    return out


def adam_flat(p, g, m, v, count, lr, beta1, beta2, eps, bc1, bc2, st=None):
    check(ops.udaseg_adam_flat(p, g, m, v, count, lr, beta1, beta2, eps,
                                        bc1, bc2, st), "adam_flat")


def fill(t, value, st=None):
    check(ops.udaseg_fill_f32(t, t.numel(), value, st), "fill_f32")


def axpy(y, x, alpha=1.0, st=None):
    check(ops.udaseg_axpy_f32(y, x, y.numel(), alpha, st),
          "axpy_f32")


def set_option(name, value):
    """Override one switch of the library's switchboard (``_lib.OPTIONS``: name -> UDASEG_OPT_* key; the environment variable of the
    same name is only its default); value -1 restores the default.  Every override bumps ``option_epoch()``."""
    key = _lib.OPTIONS[name] if isinstance(name, str) else int(name)
    check(ops.udaseg_set_option(key, int(value)), f"set_option({name})")


def get_option(name):
    key = _lib.OPTIONS[name] if isinstance(name, str) else int(name)
    return int(ops.udaseg_get_option(key))


def option_epoch():
    """Number of overrides made so far: the plans key their cached routing answers with it (engine.Plan)."""
    return int(ops.udaseg_option_epoch())


def set_generic_gather(value):
    """1: convolution kernels keep their generic gather loops; 0: uniform-tap / row-uniform loops allowed; -1: environment."""
    set_option("GENERIC_GATHER", value)


def set_f32_split(value):
    """0: the shared-source fp32 kernels stay on the fp32 matrix pipe; 1: three-term bf16 split allowed; -1: environment."""
    set_option("F32_SPLIT", value)


def prof_enable(on):
    check(ops.udaseg_prof_enable(int(on)))


def prof_reset():
    check(ops.udaseg_prof_reset())


def prof_read(family):
    ms, fl, n = _lib.C.c_double(), _lib.C.c_double(), _lib.C.c_int64()
    check(ops.udaseg_prof_read(family, _byref(ms), _byref(fl), _byref(n)), "prof_read")
    return ms.value, fl.value, n.value


def prof_records(family, max_records=4096):
    """[(ms, flops, kind, (n,hi,wi,ci,ho,wo,co,kh,kw,stride,pad))] of the recorded launches of one kernel family."""
    C = _lib.C
    ms, fl = (C.c_double * max_records)(), (C.c_double * max_records)()
    kind, desc = (C.c_int * max_records)(), (C.c_int * (11 * max_records))()
    n = ops.udaseg_prof_records(family, max_records, ms, fl, kind, desc)
    if n < 0:
        check(n, "prof_records")
    return [(ms[i], fl[i], kind[i], tuple(desc[11 * i:11 * i + 11])) for i in range(n)]


def prof_kernels():
    """[(kernel symbol, total ms, total flops, launches)] for every conv kernel instantiation."""
    lib = _lib.load()
    out = []
    for kid in range(lib.udaseg_prof_kernel_count()):
        ms, fl, n = _lib.C.c_double(), _lib.C.c_double(), _lib.C.c_int64()
        check(lib.udaseg_prof_kernel_read(kid, _byref(ms), _byref(fl), _byref(n)), "prof_kernel_read")
        out.append((lib.udaseg_prof_kernel_name(kid).decode(), ms.value, fl.value, n.value))
    return out
