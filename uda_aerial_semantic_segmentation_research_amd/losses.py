"""Loss modules of the hot path, backed by the wavefront-reduced HIP kernels in csrc/losses.hip.

* ``CrossEntropyLoss``  -- ``nn.CrossEntropyLoss()`` exactly as the reference constructs it (no arguments: mean
  reduction, no class weights, no ignore_index, no label smoothing; reference ``src/models/train.py:208``).
* ``AdversarialLoss``   -- mirror of reference ``src/models/losses.py:7-51`` (same constructor, same two methods).
  As upstream, BCE-*with-logits* is applied to whatever the discriminator returns (its sigmoid output).
"""
import torch
import torch.nn as nn

from . import kernels as K
from ._lib import load as _load
from .engine import ceil4


# data_ptr of the last dlogits buffer -> its per-class column sums (consumed once by Unet._backward_plan)
COLSUM_SIDE_TABLE = {}


def _padded_nhwc(logits):
    """[N,C,H,W] logits -> (tensor whose storage is the padded NHWC buffer [N,H,W,ldc], ldc).  Zero-copy when the
    logits come from ``Unet.forward``; otherwise one layout kernel."""
    n, c, h, w = logits.shape
    ldc = ceil4(c)
    if (logits.dtype == torch.float32 and logits.stride(1) == 1 and logits.stride(3) == ldc and logits.stride(2) == ldc * w
            and logits.stride(0) == ldc * w * h
            and logits.untyped_storage().nbytes() - 4 * logits.storage_offset() >= 4 * n * h * w * ldc):
        return logits.as_strided((n, h, w, ldc), (h * w * ldc, w * ldc, ldc, 1), logits.storage_offset()), ldc
    return K.nchw_to_nhwc(logits.float().contiguous(), ldc), ldc


class _CrossEntropyFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target):
        n, c, h, w = logits.shape
        buf, ldc = _padded_nhwc(logits.detach())
        tgt = target.contiguous()
        pixels = n * h * w
        dev = logits.device
        lse = torch.empty(pixels, device=dev, dtype=torch.float32)
        partials = torch.empty(_load().udaseg_ce_partials(), device=dev, dtype=torch.float64)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        K.ce_fwd(buf, tgt, pixels, c, ldc, lse, partials, loss)
        ctx.save_for_backward(buf, tgt, lse)
        ctx.meta = (n, c, h, w, ldc)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        buf, tgt, lse = ctx.saved_tensors
        n, c, h, w, ldc = ctx.meta
        dl = torch.empty((n, h, w, ldc), device=buf.device, dtype=torch.float32)
        g = grad_out.detach().to(torch.float32).contiguous()
        out = dl.permute(0, 3, 1, 2)[:, :c]
        if ldc <= 32:
            # per-class sums of the gradient come out of the same pass: the head conv's bias gradient (Unet's backward
            # plan picks it up from the side table instead of re-reading the 200 MB gradient)
            parts = torch.empty(_load().udaseg_ce_partials() * ldc, device=buf.device, dtype=torch.float32)
            colsum = torch.empty(ldc, device=buf.device, dtype=torch.float32)
            K.ce_bwd(buf, tgt, lse, g, n * h * w, c, ldc, dl, parts, colsum)
            COLSUM_SIDE_TABLE.clear()
            COLSUM_SIDE_TABLE[dl.data_ptr()] = colsum
        else:
            K.ce_bwd(buf, tgt, lse, g, n * h * w, c, ldc, dl)
        return out, None


class CrossEntropyLoss(nn.Module):
    """Per-pixel cross entropy over ``[N,C,H,W]`` logits and ``[N,H,W]`` int64 targets -> 0-dim loss with grad_fn."""

    def forward(self, input, target):
        if input.device.type != "cuda":
            raise RuntimeError("CrossEntropyLoss: logits must live on the GPU (no CPU path in this build)")
        if input.dim() != 4 or target.dim() != 3 or input.shape[0] != target.shape[0] or input.shape[2:] != target.shape[1:]:
            raise ValueError(f"expected logits [N,C,H,W] and target [N,H,W]; got {tuple(input.shape)} and {tuple(target.shape)}")
        if target.dtype != torch.int64:
            target = target.long()
        return _CrossEntropyFunction.apply(input, target)


class _BCEPairFunction(torch.autograd.Function):
    """loss = w_a * mean(bce_with_logits(a, label_a)) + w_b * mean(bce_with_logits(b, label_b)); b optional."""

    @staticmethod
    def forward(ctx, a, label_a, w_a, b, label_b, w_b):
        a_c = a.detach().float().contiguous()
        loss = torch.empty((), device=a.device, dtype=torch.float32)
        K.bce_logits_fwd(a_c, label_a, w_a, loss, False)
        b_c = None
        if b is not None:
            b_c = b.detach().float().contiguous()
            K.bce_logits_fwd(b_c, label_b, w_b, loss, True)
        ctx.cfg = (label_a, w_a, label_b, w_b)
        ctx.save_for_backward(a_c, b_c)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        a_c, b_c = ctx.saved_tensors
        label_a, w_a, label_b, w_b = ctx.cfg
        g = grad_out.detach().float().contiguous()
        da = torch.empty_like(a_c)
        K.bce_logits_bwd(a_c, label_a, w_a, g, da, False)
        db = None
        if b_c is not None:
            db = torch.empty_like(b_c)
            K.bce_logits_bwd(b_c, label_b, w_b, g, db, False)
        return da, None, None, db, None, None


def _need_gpu(t, who):
    if t.device.type != "cuda":
        raise RuntimeError(f"{who}: predictions must live on the GPU (no CPU path in this build)")


class AdversarialLoss:
    def __init__(self, lambda_adv=0.001):
        """Adversarial loss for domain adaptation.  lambda_adv: weight of the generator term (default 0.001)."""
        self.lambda_adv = lambda_adv

    def discriminator_loss(self, source_pred, target_pred):
        """(BCEWithLogits(source_pred, 1) + BCEWithLogits(target_pred, 0)) / 2"""
        _need_gpu(source_pred, "AdversarialLoss.discriminator_loss")
        return _BCEPairFunction.apply(source_pred, 1.0, 0.5, target_pred, 0.0, 0.5)

    def generator_loss(self, target_pred):
        """lambda_adv * BCEWithLogits(target_pred, 1)"""
        _need_gpu(target_pred, "AdversarialLoss.generator_loss")
        return _BCEPairFunction.apply(target_pred, 1.0, float(self.lambda_adv), None, 0.0, 0.0)
