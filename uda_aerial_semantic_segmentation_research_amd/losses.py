"""Loss modules of the hot path, backed by the wavefront-reduced HIP kernels in csrc/losses.hip.

* ``CrossEntropyLoss``  -- ``nn.CrossEntropyLoss()`` exactly as the reference constructs it (no arguments: mean
  reduction, no class weights, no ignore_index, no label smoothing; reference ``src/models/train.py:208``).
* ``AdversarialLoss``   -- mirror of reference ``src/models/losses.py:7-51`` (same constructor, same two methods).
  As upstream, BCE-*with-logits* is applied to whatever the discriminator returns (its sigmoid output).
* ``DiceLoss``, ``WeightedSegmentationLoss``, ``ConsistencyLoss``, ``FineTuningLoss``, ``calculate_class_weights`` --
  the rest of the reference's loss family (``src/models/losses.py:53-342``), kernels in
  csrc/losses_seg.hip: one pass over the logits per direction, softmax rows in registers.
"""
from typing import Dict, Optional
import os

import torch
import torch.nn as nn

from . import kernels as K
from ._lib import load as _load
from .engine import ceil4


# the training step's loss makes its gradient in the forward pass (one read of the logits instead of two); UDASEG_FUSE_CE=0: two passes
FUSE_CE_BACKWARD = os.environ.get("UDASEG_FUSE_CE", "1") != "0"
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
        partials = torch.empty(_load().udaseg_ce_partials(), device=dev, dtype=torch.float64)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        ctx.meta = (n, c, h, w, ldc)
        ctx.fused = None
        if FUSE_CE_BACKWARD and ldc <= 32 and ctx.needs_input_grad[0]:
            # a training step: the gradient for an upstream gradient of 1 (what loss.backward() passes) is made in the SAME pass over
            # the logits as the loss, with the head's bias gradient (column sums); backward() only scales it if it has to
            dl = torch.empty((n, h, w, ldc), device=dev, dtype=torch.float32)
            parts = torch.empty(_load().udaseg_ce_partials() * ldc, device=dev, dtype=torch.float32)
            colsum = torch.empty(ldc, device=dev, dtype=torch.float32)
            K.ce_fwd_bwd(buf, tgt, pixels, c, ldc, partials, loss, dl, parts, colsum)
            ctx.fused = [dl, colsum]
            ctx.save_for_backward(buf, tgt)
            return loss
        lse = torch.empty(pixels, device=dev, dtype=torch.float32)
        K.ce_fwd(buf, tgt, pixels, c, ldc, lse, partials, loss)
        ctx.save_for_backward(buf, tgt, lse)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        n, c, h, w, ldc = ctx.meta
        g = grad_out.detach().to(torch.float32).contiguous()
        if ctx.fused is not None:
            dl, colsum = ctx.fused
            ctx.fused = None                           # consumed (a second backward through a retained graph takes the two-pass route)
            K.scale_unless_one(dl, g, colsum)
            COLSUM_SIDE_TABLE.clear()
            COLSUM_SIDE_TABLE[dl.data_ptr()] = colsum
            return dl.permute(0, 3, 1, 2)[:, :c], None
        if len(ctx.saved_tensors) == 2:                # fused forward, second backward: recompute the log-sum-exp
            buf, tgt = ctx.saved_tensors
            lse = torch.empty(n * h * w, device=buf.device, dtype=torch.float32)
            K.ce_fwd(buf, tgt, n * h * w, c, ldc, lse, torch.empty(_load().udaseg_ce_partials(), device=buf.device, dtype=torch.float64),
                     torch.empty((), device=buf.device, dtype=torch.float32))
        else:
            buf, tgt, lse = ctx.saved_tensors
        dl = torch.empty((n, h, w, ldc), device=buf.device, dtype=torch.float32)
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


class _BCETargetFunction(torch.autograd.Function):
    """weight * mean(bce_with_logits(x, y)) with a per-sample target vector y."""

    @staticmethod
    def forward(ctx, x, y, weight):
        x_c = x.detach().float().contiguous()
        y_c = y.detach().float().contiguous()
        loss = torch.empty((), device=x.device, dtype=torch.float32)
        K.bce_logits_target_fwd(x_c, y_c, weight, loss, False)
        ctx.save_for_backward(x_c, y_c)
        ctx.weight, ctx.shape = weight, x.shape
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        x_c, y_c = ctx.saved_tensors
        dx = torch.empty_like(x_c)
        K.bce_logits_target_bwd(x_c, y_c, ctx.weight, grad_out.detach().float().contiguous(), dx, False)
        return dx.view(ctx.shape), None, None


class BCEWithLogitsLoss(nn.Module):
    """``nn.BCEWithLogitsLoss()`` as the reference constructs it (mean reduction, no weights; ``src/models/uda.py:85``)."""

    def forward(self, input, target):
        _need_gpu(input, "BCEWithLogitsLoss")
        if input.shape != target.shape:
            raise ValueError(f"Target size ({tuple(target.shape)}) must be the same as input size ({tuple(input.shape)})")
        return _BCETargetFunction.apply(input, target.to(input.device), 1.0)


class MulticlassDiceLoss(nn.Module):
    """``smp.losses.DiceLoss(mode='multiclass')`` with its defaults (from_logits, smooth 0, eps 1e-7, no ignore_index), the
    segmentation term of the reference's ``UDALoss`` (``src/models/uda.py:84``): Dice pooled over batch and pixels per
    class, classes absent from the batch's labels contribute 0, mean over all classes."""

    def __init__(self, smooth=0.0, eps=1e-7):
        super().__init__()
        self.smooth, self.eps = smooth, eps

    def forward(self, y_pred, y_true):
        y_true = _check_seg_pair("MulticlassDiceLoss", y_pred, y_true)
        return _SegLossFunction.apply(y_pred, y_true, None, 0.0, 0.0, True, float(self.smooth), 0.0, 1.0, float(self.eps), True)


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


# ------------------------------------------------------------------------------------------ Dice / focal / consistency
def _check_seg_pair(who, logits, target):
    _need_gpu(logits, who)
    if logits.dim() != 4:
        raise ValueError(f"{who}: expected predictions [B,C,H,W], got {tuple(logits.shape)}")
    if logits.shape[1] > 32:
        raise ValueError(f"{who}: at most 32 classes are supported by the HIP kernels, got {logits.shape[1]}")
    if target.dim() == 4:
        # the reference also takes one-hot [B,C,H,W] targets; the kernels work from class indices
        target = target.argmax(dim=1)
    if target.dim() != 3 or target.shape[0] != logits.shape[0] or target.shape[1:] != logits.shape[2:]:
        raise ValueError(f"{who}: targets {tuple(target.shape)} do not match predictions {tuple(logits.shape)}")
    return target.long().contiguous()


class _SegLossFunction(torch.autograd.Function):
    """loss = focal_w * focal(logits, target) + dice_w * dice(logits, target); either weight may be 0 (term skipped)."""

    @staticmethod
    def forward(ctx, logits, target, class_weights, alpha, gamma, mean, smooth, focal_w, dice_w, dice_eps=1e-7,
                dice_pooled=False):
        n, c, h, w = logits.shape
        buf, ldc = _padded_nhwc(logits.detach())
        dev = logits.device
        loss = torch.zeros((), device=dev, dtype=torch.float32)
        coef = None
        if focal_w != 0.0:
            parts = torch.empty(K.seg_partials(), device=dev, dtype=torch.float64)
            K.focal_fwd(buf, target, class_weights, alpha, gamma, n * h * w, c, ldc, mean, parts, loss)
            if focal_w != 1.0:
                loss.mul_(focal_w)
        if dice_w != 0.0:
            sums = torch.zeros(n * 3 * c, device=dev, dtype=torch.float64)
            coef = torch.empty(n * 2 * c, device=dev, dtype=torch.float32)
            dloss = torch.empty((), device=dev, dtype=torch.float32)
            K.dice_fwd(buf, target, n, h * w, c, ldc, smooth, sums, coef, dloss, dice_eps, dice_pooled)
            loss.add_(dloss, alpha=dice_w)
        ctx.save_for_backward(buf, target, class_weights, coef)
        ctx.cfg = (n, c, h, w, ldc, alpha, gamma, mean, focal_w, dice_w)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        buf, target, class_weights, coef = ctx.saved_tensors
        n, c, h, w, ldc, alpha, gamma, mean, focal_w, dice_w = ctx.cfg
        g = grad_out.detach().float().contiguous()
        dl = torch.empty((n, h, w, ldc), device=buf.device, dtype=torch.float32)
        wrote = False
        if focal_w != 0.0:
            K.focal_bwd(buf, target, class_weights, alpha, gamma, g, focal_w / (n * h * w) if mean else focal_w, n * h * w, c,
                        ldc, dl, False)
            wrote = True
        if dice_w != 0.0:
            K.dice_bwd(buf, target, coef, g, dice_w, n, h * w, c, ldc, dl, wrote)
        return (dl.permute(0, 3, 1, 2)[:, :c],) + (None,) * 10


class DiceLoss(nn.Module):
    """1 - mean over (image, class) of the soft Dice coefficient of softmax(predictions) against the labels."""

    def __init__(self, smooth=1.0):
        super().__init__()
        self.smooth = smooth

    def forward(self, predictions, targets):
        targets = _check_seg_pair("DiceLoss", predictions, targets)
        return _SegLossFunction.apply(predictions, targets, None, 0.0, 0.0, True, float(self.smooth), 0.0, 1.0)


class WeightedSegmentationLoss(nn.Module):
    """domain_weight * (focal-modulated class-weighted cross entropy + Dice)."""

    def __init__(self, num_classes: int, class_weights: Optional[torch.Tensor] = None, alpha: float = 0.25,
                 gamma: float = 2.0, reduction: str = 'mean'):
        super().__init__()
        self.num_classes = num_classes
        self.register_buffer('class_weights', torch.ones(num_classes) if class_weights is None else class_weights)
        self.alpha = alpha
        self.gamma = gamma
        self.reduction = reduction
        self.dice_loss = DiceLoss()

    def _weights_on(self, device):
        return self.class_weights.to(device=device, dtype=torch.float32).contiguous()

    def focal_loss(self, inputs, targets):
        targets = _check_seg_pair("WeightedSegmentationLoss.focal_loss", inputs, targets)
        return _SegLossFunction.apply(inputs, targets, self._weights_on(inputs.device), float(self.alpha), float(self.gamma),
                                      self.reduction == 'mean', 1.0, 1.0, 0.0)

    def forward(self, inputs, targets, domain_weight: float = 1.0):
        if inputs.dim() == 4 and inputs.shape[1] != self.num_classes:
            raise ValueError(f"WeightedSegmentationLoss: {inputs.shape[1]} channels for {self.num_classes} classes")
        targets = _check_seg_pair("WeightedSegmentationLoss", inputs, targets)
        both = _SegLossFunction.apply(inputs, targets, self._weights_on(inputs.device), float(self.alpha), float(self.gamma),
                                      self.reduction == 'mean', float(self.dice_loss.smooth), 1.0, 1.0)
        return domain_weight * both


def calculate_class_weights(dataset, num_classes: int, method: str = 'effective_samples') -> torch.Tensor:
    """Class weights from pixel counts over ``dataset`` (items ``(image, mask)``): 'effective_samples' uses
    (1-beta)/(1-beta^n) with beta=0.9999, anything else 1/n; normalised to sum to num_classes."""
    counts = torch.zeros(num_classes)
    for _, mask in dataset:
        m = torch.as_tensor(mask).reshape(-1).long()
        m = m[(m >= 0) & (m < num_classes)]
        counts += torch.bincount(m, minlength=num_classes).to(counts.dtype)
    counts = counts.clamp(min=1.0)
    if method == 'effective_samples':
        beta = 0.9999
        weights = (1.0 - beta) / (1.0 - torch.pow(beta, counts))
    else:
        weights = 1.0 / counts
    return weights / weights.sum() * num_classes


class _ConsistencyFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred1, pred2, temperature):
        n, c, h, w = pred1.shape
        z1, ldc = _padded_nhwc(pred1.detach())
        z2, _ = _padded_nhwc(pred2.detach())
        dev = pred1.device
        parts = torch.empty(K.seg_partials(), device=dev, dtype=torch.float64)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        K.consistency_fwd(z1, z2, temperature, n, n * h * w, c, ldc, parts, loss)
        ctx.save_for_backward(z1, z2)
        ctx.cfg = (n, c, h, w, ldc, temperature)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        z1, z2 = ctx.saved_tensors
        n, c, h, w, ldc, temperature = ctx.cfg
        g = grad_out.detach().float().contiguous()
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d1 = torch.empty((n, h, w, ldc), device=z1.device, dtype=torch.float32) if need1 else None
        d2 = torch.empty((n, h, w, ldc), device=z1.device, dtype=torch.float32) if need2 else None
        if need1 or need2:
            K.consistency_bwd(z1, z2, temperature, g, 1.0, n, n * h * w, c, ldc, d1, d2)
        return (d1.permute(0, 3, 1, 2)[:, :c] if need1 else None, d2.permute(0, 3, 1, 2)[:, :c] if need2 else None, None)


class ConsistencyLoss(nn.Module):
    """Symmetric KL between the temperature-softened class distributions of two predictions of the same image."""

    def __init__(self, temperature=0.5):
        super().__init__()
        self.temperature = temperature

    def forward(self, pred1, pred2):
        _need_gpu(pred1, "ConsistencyLoss")
        if pred1.dim() != 4 or pred1.shape != pred2.shape:
            raise ValueError(f"ConsistencyLoss: expected two [B,C,H,W] predictions, got {tuple(pred1.shape)} and {tuple(pred2.shape)}")
        if pred1.shape[1] > 32:
            raise ValueError("ConsistencyLoss: at most 32 classes are supported by the HIP kernels")
        return _ConsistencyFunction.apply(pred1, pred2, float(self.temperature))

    def get_similarity_matrix(self, pred1, pred2):
        """Per-pixel cosine similarity of the two softmax distributions, [B,H,W] (visualisation helper, not on the hot
        path: plain torch ops)."""
        return torch.nn.functional.cosine_similarity(torch.softmax(pred1, dim=1), torch.softmax(pred2, dim=1), dim=1)


class FineTuningLoss(nn.Module):
    """Phase-3 objective: ramped consistency + ramped domain confusion (+ Dice on labelled samples when given).

    As upstream, ``domain_weight`` enters twice: as ``AdversarialLoss(lambda_adv=domain_weight)`` and again as the
    multiplier of that term."""

    def __init__(self, consistency_weight: float = 1.0, domain_weight: float = 0.1, supervised_weight: float = 0.1,
                 rampup_length: int = 40, temperature: float = 0.5):
        super().__init__()
        self.consistency_loss = ConsistencyLoss(temperature=temperature)
        self.domain_loss = AdversarialLoss(lambda_adv=domain_weight)
        self.supervised_loss = DiceLoss()
        self.consistency_weight = consistency_weight
        self.domain_weight = domain_weight
        self.supervised_weight = supervised_weight
        self.rampup_length = rampup_length

    def rampup(self, epoch: int) -> float:
        """Linear 0 -> 1 over the first ``rampup_length`` epochs, 1 afterwards."""
        return 1.0 if epoch >= self.rampup_length else float(epoch) / self.rampup_length

    def forward(self, pred1, pred2, domain_pred, epoch: int, supervised_pred=None,
                supervised_target=None) -> Dict[str, torch.Tensor]:
        ramp = self.rampup(epoch)
        consistency = self.consistency_loss(pred1, pred2)
        domain_confusion = self.domain_loss.generator_loss(domain_pred)
        total = consistency * (self.consistency_weight * ramp) + domain_confusion * (self.domain_weight * ramp)
        supervised = torch.tensor(0.0, device=pred1.device)
        if supervised_pred is not None and supervised_target is not None:
            supervised = self.supervised_loss(supervised_pred, supervised_target)
            total = total + supervised * self.supervised_weight
        return {'total': total, 'consistency': consistency.detach(), 'domain_confusion': domain_confusion.detach(),
                'supervised': supervised.detach(), 'rampup_weight': torch.tensor(ramp)}
