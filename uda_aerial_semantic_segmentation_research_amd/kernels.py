"""Thin tensor-level wrappers over the C-ABI (one Python function per entry point of include/udaseg.h).

No autograd here and no arithmetic: these marshal ``torch.Tensor`` storage pointers, sizes and the current HIP
stream into libudaseg_hip.so.  All activations are NHWC fp32 contiguous tensors ``[N, H, W, C]`` with C % 4 == 0.
"""
import torch

from . import _lib
from ._lib import ACT_LEAKY, ACT_NONE, ConvDesc, check

_byref = _lib.C.byref


def stream():
    """hipStream_t of torch's current stream (kernels are enqueued there, asynchronously)."""
    return torch.cuda.current_stream().cuda_stream


_WORKSPACE = {}


def ensure_workspace(device, nbytes=16 << 20):
    """Hand the library its scratch buffer (once per device; PyTorch owns the memory)."""
    key = str(device)
    if key not in _WORKSPACE:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        scr = torch.zeros(256 << 10, dtype=torch.uint8, device=device)     # zeroed once; the library keeps it zeroed
        with torch.cuda.device(device):          # the library binds the buffers to the CURRENT device
            check(_lib.load().udaseg_set_workspace(buf.data_ptr(), nbytes), "set_workspace")
            check(_lib.load().udaseg_set_stats_scratch(scr.data_ptr(), scr.numel()), "set_stats_scratch")
        _WORKSPACE[key] = buf
        _WORKSPACE[key + "/stats"] = scr
    return _WORKSPACE[key]


def _ptr(t):
    return None if t is None else t.data_ptr()


def conv_desc(n, hi, wi, ci, co, k, stride, pad):
    ho = (hi + 2 * pad - k) // stride + 1
    wo = (wi + 2 * pad - k) // stride + 1
    return ConvDesc(n, hi, wi, ci, ho, wo, co, k, k, stride, pad)


def conv2d_fwd(d, x, w, bias, y, act=ACT_NONE, slope=0.0, accumulate=False, st=None):
    if x.dtype == torch.bfloat16:
        assert not accumulate
        return conv2d_fwd_bf16(d, x, w, bias, None, y, act, slope, None, st)
    check(_lib.load().udaseg_conv2d_fwd(_byref(d), x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), act, slope,
                                         int(accumulate), st if st is not None else stream()), "conv2d_fwd")


def conv2d_fwd_bnstats(d, x, w, bias, y, stats, st=None):
    if x.dtype == torch.bfloat16:
        return conv2d_fwd_bf16(d, x, w, bias, None, y, ACT_NONE, 0.0, stats, st)
    check(_lib.load().udaseg_conv2d_fwd_bnstats(_byref(d), x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(),
                                                 stats.data_ptr(), st if st is not None else stream()), "conv2d_fwd_bnstats")


def conv2d_fwd_fused(d, x, w, bias, residual, y, act=ACT_NONE, slope=0.0, st=None):
    if x.dtype == torch.bfloat16:
        return conv2d_fwd_bf16(d, x, w, bias, residual, y, act, slope, None, st)
    check(_lib.load().udaseg_conv2d_fwd_fused(_byref(d), x.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), y.data_ptr(),
                                               act, slope, st if st is not None else stream()), "conv2d_fwd_fused")


def bn_fold(w, bias, gamma, beta, running_mean, running_var, eps, w_folded, bias_folded, st=None):
    co = w.shape[0]
    check(_lib.load().udaseg_bn_fold(w.data_ptr(), _ptr(bias), gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(),
                                      running_var.data_ptr(), eps, co, w.numel() // co, w_folded.data_ptr(),
                                      bias_folded.data_ptr(), st if st is not None else stream()), "bn_fold")


def argmax_confusion(logits_base, target, pixels, classes, ldc, confusion, pred=None, st=None):
    check(_lib.load().udaseg_argmax_confusion(logits_base.data_ptr(), target.data_ptr(), pixels, classes, ldc,
                                               confusion.data_ptr(), _ptr(pred), st if st is not None else stream()),
          "argmax_confusion")


def conv2d_fwd_bf16(d, x, w, bias, residual, y, act=ACT_NONE, slope=0.0, stats=None, st=None):
    """bf16 x / w / residual; y bf16, or fp32 when its dtype says so (logits)."""
    check(_lib.load().udaseg_conv2d_fwd_bf16(_byref(d), x.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), y.data_ptr(),
                                              int(y.dtype == torch.float32), act, slope, _ptr(stats),
                                              st if st is not None else stream()), "conv2d_fwd_bf16")


def conv2d_dgrad_bf16(d, dy, w_t, dx, accumulate=False, st=None):
    check(_lib.load().udaseg_conv2d_dgrad_bf16(_byref(d), dy.data_ptr(), w_t.data_ptr(), dx.data_ptr(), int(accumulate),
                                                st if st is not None else stream()), "conv2d_dgrad_bf16")


def conv2d_wgrad_bf16(d, x, dy, dw, accumulate=False, st=None):
    check(_lib.load().udaseg_conv2d_wgrad_bf16(_byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), int(accumulate),
                                                st if st is not None else stream()), "conv2d_wgrad_bf16")


def conv2d_dgrad(d, dy, w_t, dx, accumulate=False, st=None):
    if dy.dtype == torch.bfloat16:
        return conv2d_dgrad_bf16(d, dy, w_t, dx, accumulate, st)
    check(_lib.load().udaseg_conv2d_dgrad(_byref(d), dy.data_ptr(), w_t.data_ptr(), dx.data_ptr(), int(accumulate),
                                           st if st is not None else stream()), "conv2d_dgrad")


def conv2d_dgrad_bnreduce_ok(d, dtype=torch.float32):
    if dtype == torch.bfloat16:
        return bool(_lib.load().udaseg_conv2d_dgrad_bnreduce_bf16_ok(_byref(d)))
    return dtype == torch.float32 and bool(_lib.load().udaseg_conv2d_dgrad_bnreduce_ok(_byref(d)))


def conv2d_dgrad_bnreduce(d, dy, w_t, dx, prev_y, save_mean, save_rstd, gamma, beta, act, slope, bsums, st=None):
    """dx = dgrad AND the BatchNorm-backward reductions (sum g, sum g*xhat) of the layer whose output prev_y feeds this conv."""
    fn = _lib.load().udaseg_conv2d_dgrad_bnreduce_bf16 if dy.dtype == torch.bfloat16 else _lib.load().udaseg_conv2d_dgrad_bnreduce
    check(fn(_byref(d), dy.data_ptr(), w_t.data_ptr(), dx.data_ptr(), prev_y.data_ptr(), save_mean.data_ptr(),
             save_rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act, slope, bsums.data_ptr(),
             st if st is not None else stream()), "conv2d_dgrad_bnreduce")


def conv2d_wgrad(d, x, dy, dw, accumulate=False, st=None):
    if x.dtype == torch.bfloat16:
        return conv2d_wgrad_bf16(d, x, dy, dw, accumulate, st)
    check(_lib.load().udaseg_conv2d_wgrad(_byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), int(accumulate),
                                           st if st is not None else stream()), "conv2d_wgrad")


def conv2d_fwd_upcat(d, a, skip, w, bias, y, act=ACT_NONE, slope=0.0, stats=None, st=None):
    """y = act(conv(cat([nearest_x2(a), skip], C), w) + bias) without materialising the concatenation (+ BN statistics of y)."""
    fn = _lib.load().udaseg_conv2d_fwd_upcat_bf16 if a.dtype == torch.bfloat16 else _lib.load().udaseg_conv2d_fwd_upcat
    check(fn(_byref(d), a.data_ptr(), _ptr(skip), a.shape[-1], w.data_ptr(), _ptr(bias), y.data_ptr(), act, slope, _ptr(stats),
             st if st is not None else stream()), "conv2d_fwd_upcat")


def conv2d_dgrad_split(d, dy, w_t, dx_a, dx_b, st=None):
    fn = _lib.load().udaseg_conv2d_dgrad_split_bf16 if dy.dtype == torch.bfloat16 else _lib.load().udaseg_conv2d_dgrad_split
    check(fn(_byref(d), dy.data_ptr(), w_t.data_ptr(), dx_a.data_ptr(), dx_b.data_ptr(), dx_a.shape[-1],
             st if st is not None else stream()), "conv2d_dgrad_split")


def conv2d_wgrad_part(d, src, c_off, up, dy, dw, accumulate=True, st=None):
    fn = _lib.load().udaseg_conv2d_wgrad_part_bf16 if src.dtype == torch.bfloat16 else _lib.load().udaseg_conv2d_wgrad_part
    check(fn(_byref(d), src.data_ptr(), src.shape[-1], c_off, int(up), dy.data_ptr(), dw.data_ptr(), int(accumulate),
             st if st is not None else stream()), "conv2d_wgrad_part")


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
    _channel_vecs("bn_finalize", c, gamma=gamma, beta=beta, running_mean=running_mean, running_var=running_var, save_mean=save_mean,
                  save_rstd=save_rstd, scale=scale, shift=shift)
    _channel_vecs("bn_finalize", c, f64=2 * c * bn_replicas(), sums=sums)
    check(_lib.load().udaseg_bn_finalize(sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), pixels, c, eps, momentum,
                                          _ptr(running_mean), _ptr(running_var), _ptr(save_mean), _ptr(save_rstd), scale.data_ptr(),
                                          shift.data_ptr(), st if st is not None else stream()), "bn_finalize")


def bn_bwd_apply_recompute(dz, y, fwd_scale, fwd_shift, save_mean, save_rstd, gamma, bsums, dy, dgamma, dbeta, act, slope, st=None):
    """BatchNorm backward of a layer whose activation was never written: the mask is re-evaluated from y, fwd_scale, fwd_shift."""
    c = y.shape[-1]
    _same_layout("bn_bwd_apply_recompute", y, dz=dz, dy=dy)
    _channel_vecs("bn_bwd_apply_recompute", c, fwd_scale=fwd_scale, fwd_shift=fwd_shift, save_mean=save_mean, save_rstd=save_rstd,
                  gamma=gamma, dgamma=dgamma, dbeta=dbeta)
    _channel_vecs("bn_bwd_apply_recompute", c, f64=2 * c * bn_replicas(), bsums=bsums)
    if y.dtype != torch.bfloat16:
        raise ValueError("bn_bwd_apply_recompute: bf16 storage only")
    check(_lib.load().udaseg_bn_bwd_apply_recompute_bf16(dz.data_ptr(), y.data_ptr(), fwd_scale.data_ptr(), fwd_shift.data_ptr(),
                                                          save_mean.data_ptr(), save_rstd.data_ptr(), gamma.data_ptr(),
                                                          bsums.data_ptr(), dy.data_ptr(), _ptr(dgamma), _ptr(dbeta), y.numel() // c,
                                                          c, act, slope, st if st is not None else stream()),
          "bn_bwd_apply_recompute_bf16")


def conv2d_wgrad_halo_ok(d, up_ca=0, f32=False):
    if f32:
        return bool(_lib.load().udaseg_conv2d_wgrad_halo_f32x3_ok(_byref(d), up_ca))
    return bool(_lib.load().udaseg_conv2d_wgrad_halo_bf16_ok(_byref(d), up_ca))


def conv2d_wgrad_halo(d, x, skip, dy, dw, up=False, st=None):
    """dW += weight gradient of a stride-1 3x3 layer (channel counts multiples of 64) on the halo-resident kernel: bf16 tensors,
    or fp32 tensors with the exact three-term split.  up: x is the half-resolution source of a fused decoder input, skip the
    other one."""
    fn, name = ((_lib.load().udaseg_conv2d_wgrad_halo_f32x3, "conv2d_wgrad_halo_f32x3") if x.dtype == torch.float32
                else (_lib.load().udaseg_conv2d_wgrad_halo_bf16, "conv2d_wgrad_halo_bf16"))
    check(fn(_byref(d), x.data_ptr(), _ptr(skip), x.shape[-1] if up else 0, dy.data_ptr(), dw.data_ptr(),
             st if st is not None else stream()), name)


def conv2d_wgrad_bnin(d, y_prev, in_scale, in_shift, in_act, in_slope, dy, dw, accumulate=False, st=None):
    """Weight gradient whose gathered operand is act(fma(y_prev, in_scale, in_shift)) (never written), bf16."""
    check(_lib.load().udaseg_conv2d_wgrad_bnin_bf16(_byref(d), y_prev.data_ptr(), in_scale.data_ptr(), in_shift.data_ptr(), in_act,
                                                     in_slope, dy.data_ptr(), dw.data_ptr(), int(accumulate),
                                                     st if st is not None else stream()), "conv2d_wgrad_bnin_bf16")


def frag_elems(n_out, k_in, ks):
    """bf16 elements of the MFMA-fragment packing of a convolution with n_out produced / k_in gathered channels, ks x ks window."""
    return int(_lib.load().udaseg_frag_elems(n_out, k_in, ks))


def pack_frag_batched(w, wt, packed, table, st=None):
    """Fragment-pack every listed convolution in one launch (table rows: mode, src offset, dst offset, N, K, ks; int32 x 6).
    bf16 sources: one plane (csrc/conv_halo_bf16.hip); fp32 sources: the three split planes (csrc/conv_halo_f32x3.hip)."""
    src = w if w is not None else wt
    if src.dtype == torch.float32:
        check(_lib.load().udaseg_pack_frag_batched_f32x3(_ptr(w), _ptr(wt), packed.data_ptr(), table.data_ptr(), table.shape[0],
                                                          st if st is not None else stream()), "pack_frag_batched_f32x3")
        return
    check(_lib.load().udaseg_pack_frag_batched_bf16(_ptr(w), _ptr(wt), packed.data_ptr(), table.data_ptr(), table.shape[0],
                                                     st if st is not None else stream()), "pack_frag_batched_bf16")


def conv_frag_ok(d, dgrad=False, up_ca=0, f32=False):
    if f32:
        return bool(_lib.load().udaseg_conv_f32x3_ok(_byref(d), int(dgrad), up_ca))
    return bool(_lib.load().udaseg_conv_frag_ok(_byref(d), int(dgrad), up_ca))


def conv_frag_preferred(d, dgrad=False, up_ca=0, f32=False):
    """Supported AND expected to beat the shared implicit-GEMM kernel for this shape (the library's measured heuristic)."""
    if f32:
        return bool(_lib.load().udaseg_conv_f32x3_preferred(_byref(d), int(dgrad), up_ca))
    return bool(_lib.load().udaseg_conv_frag_preferred(_byref(d), int(dgrad), up_ca))


def conv2d_fwd_frag(d, x, skip, wfrag, bias, y, act=ACT_NONE, slope=0.0, stats=None, in_scale=None, in_shift=None, in_act=ACT_NONE,
                    in_slope=0.0, up=False, st=None):
    """Halo-resident forward convolution on the bf16 matrix pipe: bf16 tensors (csrc/conv_halo_bf16.hip) or fp32 tensors with
    the three-term split (csrc/conv_halo_f32x3.hip).  up: x is the half-resolution tensor of a fused decoder input."""
    if x.dtype == torch.float32:
        assert in_scale is None and y.dtype == torch.float32
        check(_lib.load().udaseg_conv2d_fwd_f32x3(_byref(d), x.data_ptr(), _ptr(skip), x.shape[-1] if up else 0, wfrag.data_ptr(),
                                                   _ptr(bias), y.data_ptr(), act, slope, _ptr(stats),
                                                   st if st is not None else stream()), "conv2d_fwd_f32x3")
        return
    check(_lib.load().udaseg_conv2d_fwd_frag_bf16(_byref(d), x.data_ptr(), _ptr(skip), x.shape[-1] if up else 0, wfrag.data_ptr(),
                                                   _ptr(bias), _ptr(in_scale), _ptr(in_shift), in_act, in_slope, y.data_ptr(),
                                                   int(y.dtype == torch.float32), act, slope, _ptr(stats),
                                                   st if st is not None else stream()), "conv2d_fwd_frag_bf16")


def conv2d_dgrad_frag(d, dy, wfrag_t, dx, dx2=None, bn=None, accumulate=False, st=None):
    """Halo-resident data gradient (bf16 tensors, or fp32 tensors with the three-term split).  dx2: second destination of a
    split gradient (channels [dx.shape[-1], ci)).
    bn = (prev_y, save_mean, save_rstd, gamma, beta, act, slope, bsums): BatchNorm-backward reductions of the layer behind."""
    py, mu, rs, ga, be, act, slope, bs = bn if bn is not None else (None, None, None, None, None, ACT_NONE, 0.0, None)
    fn, name = ((_lib.load().udaseg_conv2d_dgrad_f32x3, "conv2d_dgrad_f32x3") if dy.dtype == torch.float32
                else (_lib.load().udaseg_conv2d_dgrad_frag_bf16, "conv2d_dgrad_frag_bf16"))
    check(fn(_byref(d), dy.data_ptr(), wfrag_t.data_ptr(), dx.data_ptr(), _ptr(dx2), dx.shape[-1] if dx2 is not None else 0,
             _ptr(py), _ptr(mu), _ptr(rs), _ptr(ga), _ptr(be), act, slope, _ptr(bs), int(accumulate),
             st if st is not None else stream()), name)


def pack_dgrad_weights(d, w, w_t, st=None):
    check(_lib.load().udaseg_pack_dgrad_weights(_byref(d), w.data_ptr(), w_t.data_ptr(),
                                                 st if st is not None else stream()), "pack_dgrad_weights")


def cast_to_bf16(x, out=None, st=None):
    """fp32 -> bf16 (round to nearest even), flat; numel must be a multiple of 8."""
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(_lib.load().udaseg_cast_f32_to_bf16(x.data_ptr(), out.data_ptr(), x.numel(), st if st is not None else stream()),
          "cast_f32_to_bf16")
    return out


def pack_dgrad_batched(arena, packed, table, st=None):
    if packed.dtype == torch.bfloat16:
        check(_lib.load().udaseg_pack_dgrad_batched_bf16(arena.data_ptr(), packed.data_ptr(), table.data_ptr(), table.shape[0],
                                                          st if st is not None else stream()), "pack_dgrad_batched_bf16")
        return
    check(_lib.load().udaseg_pack_dgrad_batched(arena.data_ptr(), packed.data_ptr(), table.data_ptr(), table.shape[0],
                                                 st if st is not None else stream()), "pack_dgrad_batched")


def conv_flops(d):
    return _lib.load().udaseg_conv_flops(_byref(d))


def nchw_to_nhwc(x, cpad=None, st=None, dtype=torch.float32):
    """[N,C,H,W] contiguous fp32 -> [N,H,W,cpad] (zero-padded channels), fp32 or bf16."""
    n, c, h, w = x.shape
    if not x.is_contiguous():
        x = x.contiguous()
    if dtype == torch.bfloat16:
        cpad = cpad or ((c + 7) // 8) * 8
        y = torch.empty((n, h, w, cpad), device=x.device, dtype=torch.bfloat16)
        check(_lib.load().udaseg_nchw_to_nhwc_bf16(x.data_ptr(), y.data_ptr(), n, c, h, w, cpad,
                                                    st if st is not None else stream()), "nchw_to_nhwc_bf16")
        return y
    cpad = cpad or ((c + 3) // 4) * 4
    y = torch.empty((n, h, w, cpad), device=x.device, dtype=torch.float32)
    check(_lib.load().udaseg_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), n, c, h, w, cpad,
                                           st if st is not None else stream()), "nchw_to_nhwc")
    return y


def _same_layout(who, y, **others):
    """The element-wise entry points take ONE extent (pixels, c) for all their activation operands and raw pointers for each:
    a tensor of another dtype or size would be read / written past its end by the kernel (a GPU memory fault, not an error
    code -- the C-ABI has no per-pointer extents).  The binding therefore refuses operands that do not share y's dtype, element
    count and device, or are not contiguous.  (Round 2's tools/bn_bandwidth.py fault: profiles/r02_bn_bandwidth.txt.)"""
    n, dt, dev = y.numel(), y.dtype, y.device
    if not y.is_contiguous():
        raise ValueError(f"{who}: y must be contiguous")
    for name, t in others.items():
        if t is None:
            continue
        if t.dtype is not dt or t.numel() != n or t.device != dev or not t.is_contiguous():
            raise ValueError(f"{who}: {name} must match y (dtype {dt}, {n} elements, contiguous, {dev}); got dtype {t.dtype}, "
                             f"{t.numel()} elements on {t.device}")


def _channel_vecs(who, c, f64=0, **vecs):
    """Per-channel operands: fp32 vectors of >= c elements (f64 > 0: f64 accumulators of >= f64 elements)."""
    for name, t in vecs.items():
        if t is None:
            continue
        want = torch.float64 if f64 else torch.float32
        need = f64 if f64 else c
        if t.dtype is not want or t.numel() < need or not t.is_contiguous():
            raise ValueError(f"{who}: {name} must be a contiguous {want} tensor of >= {need} elements; got {t.dtype}, {t.numel()}")


_BN_R = None


def bn_replicas():
    global _BN_R
    if _BN_R is None:
        _BN_R = _lib.load().udaseg_bn_replicas()
    return _BN_R


def bn_stats(y, sums, st=None):
    c = y.shape[-1]
    if y.dtype not in (torch.float32, torch.bfloat16) or not y.is_contiguous():
        raise ValueError(f"bn_stats: contiguous fp32 or bf16 tensor expected, got {y.dtype}")
    _channel_vecs("bn_stats", c, f64=2 * c * bn_replicas(), sums=sums)
    fn = _lib.load().udaseg_bn_stats_bf16 if y.dtype == torch.bfloat16 else _lib.load().udaseg_bn_stats
    check(fn(y.data_ptr(), y.numel() // c, c, sums.data_ptr(), st if st is not None else stream()), "bn_stats")


def bn_apply(y, sums, gamma, beta, residual, z, eps, momentum, running_mean, running_var, save_mean, save_rstd, act, slope,
             st=None):
    c = y.shape[-1]
    _same_layout("bn_apply", y, residual=residual, z=z)
    _channel_vecs("bn_apply", c, gamma=gamma, beta=beta, running_mean=running_mean, running_var=running_var,
                  save_mean=save_mean, save_rstd=save_rstd)
    _channel_vecs("bn_apply", c, f64=2 * c * bn_replicas(), sums=sums)
    if y.dtype == torch.bfloat16:
        check(_lib.load().udaseg_bn_apply_bf16(y.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(residual),
                                                z.data_ptr(), y.numel() // c, c, eps, momentum, _ptr(running_mean),
                                                _ptr(running_var), _ptr(save_mean), _ptr(save_rstd), act, slope,
                                                st if st is not None else stream()), "bn_apply_bf16")
        return
    check(_lib.load().udaseg_bn_apply(y.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(residual),
                                       z.data_ptr(), y.numel() // c, c, eps, momentum, _ptr(running_mean), _ptr(running_var),
                                       _ptr(save_mean), _ptr(save_rstd), act, slope,
                                       st if st is not None else stream()), "bn_apply")


def bn_apply_eval(y, gamma, beta, running_mean, running_var, residual, z, eps, act, slope, st=None):
    c = y.shape[-1]
    check(_lib.load().udaseg_bn_apply_eval(y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(),
                                            running_var.data_ptr(), _ptr(residual), z.data_ptr(), y.numel() // c, c, eps, act,
                                            slope, st if st is not None else stream()), "bn_apply_eval")


def bn_bwd_reduce(dz, z, y, save_mean, save_rstd, bsums, act, slope, st=None, gamma=None, beta=None):
    """z=None (fp32 only, layers without a residual input): the activation's argument is re-evaluated from y, gamma, beta."""
    c = y.shape[-1]
    _same_layout("bn_bwd_reduce", y, dz=dz, z=z)
    _channel_vecs("bn_bwd_reduce", c, save_mean=save_mean, save_rstd=save_rstd, gamma=gamma, beta=beta)
    _channel_vecs("bn_bwd_reduce", c, f64=2 * c * bn_replicas(), bsums=bsums)
    if y.dtype == torch.bfloat16:
        check(_lib.load().udaseg_bn_bwd_reduce_bf16(dz.data_ptr(), _ptr(z), y.data_ptr(), save_mean.data_ptr(),
                                                     save_rstd.data_ptr(), y.numel() // c, c, bsums.data_ptr(), act, slope,
                                                     st if st is not None else stream()), "bn_bwd_reduce_bf16")
        return
    check(_lib.load().udaseg_bn_bwd_reduce(dz.data_ptr(), _ptr(z), y.data_ptr(), save_mean.data_ptr(), save_rstd.data_ptr(),
                                            _ptr(gamma), _ptr(beta), y.numel() // c, c, bsums.data_ptr(), act, slope,
                                            st if st is not None else stream()), "bn_bwd_reduce")


def bn_bwd_apply(dz, z, y, save_mean, save_rstd, gamma, bsums, dy, dres, dgamma, dbeta, act, slope, accumulate_dy=False,
                 accumulate_dres=False, accumulate_param=False, st=None, beta=None):
    c = y.shape[-1]
    _same_layout("bn_bwd_apply", y, dz=dz, z=z, dy=dy, dres=dres)
    _channel_vecs("bn_bwd_apply", c, save_mean=save_mean, save_rstd=save_rstd, gamma=gamma, beta=beta, dgamma=dgamma, dbeta=dbeta)
    _channel_vecs("bn_bwd_apply", c, f64=2 * c * bn_replicas(), bsums=bsums)
    if y.dtype == torch.bfloat16:
        check(_lib.load().udaseg_bn_bwd_apply_bf16(dz.data_ptr(), _ptr(z), y.data_ptr(), save_mean.data_ptr(),
                                                    save_rstd.data_ptr(), gamma.data_ptr(), bsums.data_ptr(), dy.data_ptr(),
                                                    _ptr(dres), _ptr(dgamma), _ptr(dbeta), y.numel() // c, c, act, slope,
                                                    int(accumulate_dy), int(accumulate_dres), int(accumulate_param),
                                                    st if st is not None else stream()), "bn_bwd_apply_bf16")
        return
    check(_lib.load().udaseg_bn_bwd_apply(dz.data_ptr(), _ptr(z), y.data_ptr(), save_mean.data_ptr(), save_rstd.data_ptr(),
                                           gamma.data_ptr(), _ptr(beta), bsums.data_ptr(), dy.data_ptr(), _ptr(dres), _ptr(dgamma),
                                           _ptr(dbeta), y.numel() // c, c, act, slope, int(accumulate_dy),
                                           int(accumulate_dres), int(accumulate_param),
                                           st if st is not None else stream()), "bn_bwd_apply")


def act_bwd(dz, z, dy, act, slope, st=None):
    if dz.dtype == torch.bfloat16:
        check(_lib.load().udaseg_act_bwd_bf16(dz.data_ptr(), z.data_ptr(), dy.data_ptr(), dz.numel(), act, slope,
                                               st if st is not None else stream()), "act_bwd_bf16")
        return
    check(_lib.load().udaseg_act_bwd(dz.data_ptr(), z.data_ptr(), dy.data_ptr(), dz.numel(), act, slope,
                                      st if st is not None else stream()), "act_bwd")


_CHSUM_SCRATCH = {}          # (device index, stream handle) -> persistent fp32 scratch for channel_sum's partial sums


def channel_sum(x, out, accumulate=False, st=None):
    """out[c] (+)= sum over pixels.  Large inputs reduce through 16 replicas of ``out`` in a scratch that belongs to the
    stream the call runs on (kept per stream: calls on one stream are ordered, calls on different streams never share it)."""
    c = x.shape[-1]
    st = st if st is not None else stream()
    key = (x.device.index, st)
    ws = _CHSUM_SCRATCH.get(key)
    need = 16 * c
    if ws is None or ws.numel() < need:
        ws = _CHSUM_SCRATCH[key] = torch.empty(max(need, 16 * 2048), dtype=torch.float32, device=x.device)
    fn = _lib.load().udaseg_channel_sum_bf16_ws if x.dtype == torch.bfloat16 else _lib.load().udaseg_channel_sum_ws
    check(fn(x.data_ptr(), x.numel() // c, c, out.data_ptr(), int(accumulate), ws.data_ptr(), ws.numel() * 4, st), "channel_sum")


def maxpool_fwd(x, st=None):
    n, h, w, c = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((n, ho, wo, c), device=x.device, dtype=x.dtype)
    idx = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8)
    if x.dtype == torch.bfloat16:
        check(_lib.load().udaseg_maxpool3x3s2_fwd_bf16(x.data_ptr(), y.data_ptr(), idx.data_ptr(), n, h, w, c,
                                                        st if st is not None else stream()), "maxpool_fwd_bf16")
        return y, idx
    check(_lib.load().udaseg_maxpool3x3s2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr(), n, h, w, c,
                                               st if st is not None else stream()), "maxpool_fwd")
    return y, idx


def maxpool_bwd(dy, idx, dx, accumulate=False, st=None):
    n, h, w, c = dx.shape
    if dx.dtype == torch.bfloat16:
        check(_lib.load().udaseg_maxpool3x3s2_bwd_bf16(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), n, h, w, c, int(accumulate),
                                                        st if st is not None else stream()), "maxpool_bwd_bf16")
        return
    check(_lib.load().udaseg_maxpool3x3s2_bwd(dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), n, h, w, c, int(accumulate),
                                               st if st is not None else stream()), "maxpool_bwd")


def upsample2x_concat_fwd(a, skip, st=None):
    n, h, w, ca = a.shape
    cb = 0 if skip is None else skip.shape[-1]
    out = torch.empty((n, 2 * h, 2 * w, ca + cb), device=a.device, dtype=a.dtype)
    if a.dtype == torch.bfloat16:      # pure data movement in 16-byte vectors: 8 bf16 channels == 4 fp32 "channels"
        ca, cb = ca // 2, cb // 2
    check(_lib.load().udaseg_upsample2x_concat_fwd(a.data_ptr(), _ptr(skip), out.data_ptr(), n, h, w, ca, cb,
                                                    st if st is not None else stream()), "upsample2x_concat_fwd")
    return out


def upsample2x_concat_bwd(dout, da, dskip, ca, cb, accumulate_da=False, accumulate_dskip=False, st=None):
    n, h2, w2, _ = dout.shape
    if dout.dtype == torch.bfloat16:
        check(_lib.load().udaseg_upsample2x_concat_bwd_bf16(dout.data_ptr(), _ptr(da), _ptr(dskip), n, h2 // 2, w2 // 2, ca, cb,
                                                             int(accumulate_da), int(accumulate_dskip),
                                                             st if st is not None else stream()), "upsample2x_concat_bwd_bf16")
        return
    check(_lib.load().udaseg_upsample2x_concat_bwd(dout.data_ptr(), _ptr(da), _ptr(dskip), n, h2 // 2, w2 // 2, ca, cb,
                                                    int(accumulate_da), int(accumulate_dskip),
                                                    st if st is not None else stream()), "upsample2x_concat_bwd")


def upsample2x_bilinear_concat_fwd(a, skip, st=None):
    """cat(interpolate(a, scale_factor=2, mode='bilinear', align_corners=False), skip) on NHWC tensors (fp32 or bf16)."""
    n, h, w, ca = a.shape
    cb = 0 if skip is None else skip.shape[-1]
    out = torch.empty((n, 2 * h, 2 * w, ca + cb), device=a.device, dtype=a.dtype)
    check(_lib.load().udaseg_upsample2x_bilinear_concat_fwd(a.data_ptr(), _ptr(skip), out.data_ptr(), n, h, w, ca, cb,
                                                             int(a.dtype == torch.bfloat16), st if st is not None else stream()),
          "upsample2x_bilinear_concat_fwd")
    return out


def upsample2x_bilinear_concat_bwd(dout, da, dskip, ca, cb, accumulate_da=False, accumulate_dskip=False, st=None):
    n, h2, w2, _ = dout.shape
    check(_lib.load().udaseg_upsample2x_bilinear_concat_bwd(dout.data_ptr(), _ptr(da), _ptr(dskip), n, h2 // 2, w2 // 2, ca, cb,
                                                             int(accumulate_da), int(accumulate_dskip),
                                                             int(dout.dtype == torch.bfloat16), st if st is not None else stream()),
          "upsample2x_bilinear_concat_bwd")


def ce_fwd(logits_base, target, pixels, classes, ldc, lse, partials, loss, st=None):
    check(_lib.load().udaseg_ce_fwd(logits_base.data_ptr(), target.data_ptr(), pixels, classes, ldc, lse.data_ptr(),
                                     partials.data_ptr(), loss.data_ptr(), st if st is not None else stream()), "ce_fwd")


def ce_bwd(logits_base, target, lse, grad_out, pixels, classes, ldc, dlogits, colsum_partials=None, colsum=None, st=None):
    check(_lib.load().udaseg_ce_bwd(logits_base.data_ptr(), target.data_ptr(), lse.data_ptr(), _ptr(grad_out), pixels, classes,
                                     ldc, dlogits.data_ptr(), _ptr(colsum_partials), _ptr(colsum),
                                     st if st is not None else stream()), "ce_bwd")


def seg_partials():
    return _lib.load().udaseg_seg_partials()


def dice_fwd(logits_base, target, batch, pix_per_image, classes, ldc, smooth, sums, coef, loss, eps=1e-7, pooled=False, st=None):
    check(_lib.load().udaseg_dice_fwd(logits_base.data_ptr(), target.data_ptr(), batch, pix_per_image, classes, ldc,
                                       float(smooth), float(eps), int(pooled), sums.data_ptr(), coef.data_ptr(),
                                       loss.data_ptr(), st if st is not None else stream()), "dice_fwd")


def dice_bwd(logits_base, target, coef, grad_out, weight, batch, pix_per_image, classes, ldc, dlogits, accumulate=False,
             st=None):
    check(_lib.load().udaseg_dice_bwd(logits_base.data_ptr(), target.data_ptr(), coef.data_ptr(), _ptr(grad_out),
                                       float(weight), batch, pix_per_image, classes, ldc, dlogits.data_ptr(),
                                       int(accumulate), st if st is not None else stream()), "dice_bwd")


def focal_fwd(logits_base, target, class_weights, alpha, gamma, pixels, classes, ldc, mean, partials, loss, accumulate=False,
              st=None):
    check(_lib.load().udaseg_focal_fwd(logits_base.data_ptr(), target.data_ptr(), _ptr(class_weights), float(alpha),
                                        float(gamma), pixels, classes, ldc, int(mean), partials.data_ptr(), loss.data_ptr(),
                                        int(accumulate), st if st is not None else stream()), "focal_fwd")


def focal_bwd(logits_base, target, class_weights, alpha, gamma, grad_out, weight, pixels, classes, ldc, dlogits,
              accumulate=False, st=None):
    check(_lib.load().udaseg_focal_bwd(logits_base.data_ptr(), target.data_ptr(), _ptr(class_weights), float(alpha),
                                        float(gamma), _ptr(grad_out), float(weight), pixels, classes, ldc,
                                        dlogits.data_ptr(), int(accumulate), st if st is not None else stream()), "focal_bwd")


def consistency_fwd(z1, z2, temperature, batch, pixels, classes, ldc, partials, loss, st=None):
    check(_lib.load().udaseg_consistency_fwd(z1.data_ptr(), z2.data_ptr(), float(temperature), batch, pixels, classes, ldc,
                                              partials.data_ptr(), loss.data_ptr(), st if st is not None else stream()),
          "consistency_fwd")


def consistency_bwd(z1, z2, temperature, grad_out, weight, batch, pixels, classes, ldc, d1, d2, accumulate=False, st=None):
    check(_lib.load().udaseg_consistency_bwd(z1.data_ptr(), z2.data_ptr(), float(temperature), _ptr(grad_out), float(weight),
                                              batch, pixels, classes, ldc, _ptr(d1), _ptr(d2), int(accumulate),
                                              st if st is not None else stream()), "consistency_bwd")


def gap_linear_sigmoid_fwd(z, w, b, st=None):
    n, h, wd, c = z.shape
    hw = h * wd
    splits = _lib.load().udaseg_gap_splits(hw)
    partial = torch.empty((n, splits, c), device=z.device, dtype=torch.float32)
    pooled = torch.empty((n, c), device=z.device, dtype=torch.float32)
    p = torch.empty((n, 1), device=z.device, dtype=torch.float32)
    if z.dtype == torch.bfloat16:
        sv = st if st is not None else stream()
        check(_lib.load().udaseg_gap_partial_bf16(z.data_ptr(), partial.data_ptr(), n, hw, c, sv), "gap_partial_bf16")
        check(_lib.load().udaseg_gap_finish(partial.data_ptr(), w.data_ptr(), b.data_ptr(), pooled.data_ptr(), p.data_ptr(), n, hw,
                                             c, sv), "gap_finish")
        return p, pooled
    check(_lib.load().udaseg_gap_linear_sigmoid_fwd(z.data_ptr(), w.data_ptr(), b.data_ptr(), partial.data_ptr(),
                                                     pooled.data_ptr(), p.data_ptr(), n, hw, c,
                                                     st if st is not None else stream()), "gap_linear_sigmoid_fwd")
    return p, pooled


def gap_linear_sigmoid_bwd(dp, p, pooled, w, dz, dw, db, accumulate_param=False, st=None):
    n, h, wd, c = dz.shape
    if dz.dtype == torch.bfloat16:
        sv = st if st is not None else stream()
        check(_lib.load().udaseg_gap_bwd_broadcast_bf16(dp.data_ptr(), p.data_ptr(), w.data_ptr(), dz.data_ptr(), n, h * wd, c, sv),
              "gap_bwd_broadcast_bf16")
        check(_lib.load().udaseg_gap_bwd_param(dp.data_ptr(), p.data_ptr(), pooled.data_ptr(), dw.data_ptr(), db.data_ptr(), n, c,
                                                int(accumulate_param), sv), "gap_bwd_param")
        return
    check(_lib.load().udaseg_gap_linear_sigmoid_bwd(dp.data_ptr(), p.data_ptr(), pooled.data_ptr(), w.data_ptr(), dz.data_ptr(),
                                                     dw.data_ptr(), db.data_ptr(), n, h * wd, c, int(accumulate_param),
                                                     st if st is not None else stream()), "gap_linear_sigmoid_bwd")


def bce_logits_fwd(x, label, weight, loss, accumulate=False, st=None):
    check(_lib.load().udaseg_bce_logits_fwd(x.data_ptr(), x.numel(), label, weight, loss.data_ptr(), int(accumulate),
                                             st if st is not None else stream()), "bce_logits_fwd")


def bce_logits_bwd(x, label, weight, grad_out, dx, accumulate=False, st=None):
    check(_lib.load().udaseg_bce_logits_bwd(x.data_ptr(), x.numel(), label, weight, _ptr(grad_out), dx.data_ptr(),
                                             int(accumulate), st if st is not None else stream()), "bce_logits_bwd")


def gap_linear_fwd(z, w, b, st=None):
    """logit[n] = dot(mean over pixels of z[n], w) + b for NHWC z; returns (logit [n], pooled [n, c])."""
    n, h, wd, c = z.shape
    hw = h * wd
    splits = _lib.load().udaseg_gap_splits(hw)
    partial = torch.empty((n, splits, c), device=z.device, dtype=torch.float32)
    pooled = torch.empty((n, c), device=z.device, dtype=torch.float32)
    logit = torch.empty(n, device=z.device, dtype=torch.float32)
    check(_lib.load().udaseg_gap_linear_fwd(z.data_ptr(), w.data_ptr(), b.data_ptr(), partial.data_ptr(), pooled.data_ptr(),
                                             logit.data_ptr(), n, hw, c, st if st is not None else stream()), "gap_linear_fwd")
    return logit, pooled


def gap_linear_bwd(dlogit, pooled, w, dz, dw, db, accumulate_param=False, st=None):
    n, h, wd, c = dz.shape
    check(_lib.load().udaseg_gap_linear_bwd(dlogit.data_ptr(), pooled.data_ptr(), w.data_ptr(), dz.data_ptr(), dw.data_ptr(),
                                             db.data_ptr(), n, h * wd, c, int(accumulate_param),
                                             st if st is not None else stream()), "gap_linear_bwd")


def bce_logits_target_fwd(x, target, weight, loss, accumulate=False, st=None):
    check(_lib.load().udaseg_bce_logits_target_fwd(x.data_ptr(), target.data_ptr(), x.numel(), float(weight), loss.data_ptr(),
                                                    int(accumulate), st if st is not None else stream()), "bce_logits_target_fwd")


def bce_logits_target_bwd(x, target, weight, grad_out, dx, accumulate=False, st=None):
    check(_lib.load().udaseg_bce_logits_target_bwd(x.data_ptr(), target.data_ptr(), x.numel(), float(weight), _ptr(grad_out),
                                                    dx.data_ptr(), int(accumulate), st if st is not None else stream()),
          "bce_logits_target_bwd")


def scale(x, alpha, out=None, st=None):
    """out = alpha * x (dense fp32)."""
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().udaseg_scale_f32(x.data_ptr(), out.data_ptr(), x.numel(), float(alpha),
                                        st if st is not None else stream()), "scale_f32")
    return out


def adam_flat(p, g, m, v, count, lr, beta1, beta2, eps, bc1, bc2, st=None):
    check(_lib.load().udaseg_adam_flat(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), count, lr, beta1, beta2, eps,
                                        bc1, bc2, st if st is not None else stream()), "adam_flat")


def fill(t, value, st=None):
    check(_lib.load().udaseg_fill_f32(t.data_ptr(), t.numel(), value, st if st is not None else stream()), "fill_f32")


def axpy(y, x, alpha=1.0, st=None):
    check(_lib.load().udaseg_axpy_f32(y.data_ptr(), x.data_ptr(), y.numel(), alpha, st if st is not None else stream()),
          "axpy_f32")


def set_generic_gather(value):
    """1: convolution kernels keep their generic gather loops; 0: uniform-tap / row-uniform loops allowed; -1: environment."""
    check(_lib.load().udaseg_set_option(0, int(value)), "set_option")


def prof_enable(on):
    check(_lib.load().udaseg_prof_enable(int(on)))


def prof_reset():
    check(_lib.load().udaseg_prof_reset())


def prof_read(family):
    ms, fl, n = _lib.C.c_double(), _lib.C.c_double(), _lib.C.c_int64()
    check(_lib.load().udaseg_prof_read(family, _byref(ms), _byref(fl), _byref(n)), "prof_read")
    return ms.value, fl.value, n.value


def prof_records(family, max_records=4096):
    """[(ms, flops, kind, (n,hi,wi,ci,ho,wo,co,kh,kw,stride,pad))] of the recorded launches of one kernel family."""
    C = _lib.C
    ms, fl = (C.c_double * max_records)(), (C.c_double * max_records)()
    kind, desc = (C.c_int * max_records)(), (C.c_int * (11 * max_records))()
    n = _lib.load().udaseg_prof_records(family, max_records, ms, fl, kind, desc)
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
