"""``DomainAdaptationMetrics`` -- mirror of reference ``src/models/metrics.py:5-73`` (running domain accuracy and the
entropy of ``sigmoid(pred)``; the reference re-applies the sigmoid to probabilities), plus the per-batch
segmentation metrics of ``SegmentationTrainer.calculate_metrics`` (reference ``src/models/train.py:225-243``) and
``SegmentationMetrics`` -- mirror of reference ``src/analysis/metrics.py:5-67`` on the device-side confusion matrix.

These are logging side-cars, not the hot path: plain torch ops with ONE device->host transfer per call (the reference
issues >= 25 ``.item()`` syncs per step).
"""
import torch
import torch.nn.functional as F


class DomainAdaptationMetrics:
    """Track metrics for domain adaptation training."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.source_correct = 0
        self.source_total = 0
        self.target_correct = 0
        self.target_total = 0
        self.domain_entropy_sum = 0.0
        self.feature_alignment_sum = 0.0
        self.n_batches = 0

    def update(self, source_pred, target_pred, source_features=None, target_features=None):
        source_pred, target_pred = source_pred.detach(), target_pred.detach()
        p = torch.sigmoid(torch.cat([source_pred, target_pred], dim=0))
        ent = (-p * torch.log(p + 1e-10) - (1 - p) * torch.log(1 - p + 1e-10)).mean()
        packed = torch.stack([(source_pred >= 0.5).sum().float(), (target_pred < 0.5).sum().float(), ent]).cpu()
        self.source_correct += int(packed[0])
        self.source_total += source_pred.size(0)
        self.target_correct += int(packed[1])
        self.target_total += target_pred.size(0)
        self.domain_entropy_sum += float(packed[2])
        if source_features is not None and target_features is not None:
            a = F.normalize(source_features.mean(0), dim=0)
            b = F.normalize(target_features.mean(0), dim=0)
            self.feature_alignment_sum += F.cosine_similarity(a, b, dim=0).item()
        self.n_batches += 1

    def update_domain_accuracy(self, source_pred, target_pred):
        self.source_correct += int((source_pred >= 0.5).sum().item())
        self.source_total += source_pred.size(0)
        self.target_correct += int((target_pred < 0.5).sum().item())
        self.target_total += target_pred.size(0)

    def get_metrics(self):
        return {
            "source_domain_acc": f"{self.source_correct / max(self.source_total, 1):.4f}",
            "target_domain_acc": f"{self.target_correct / max(self.target_total, 1):.4f}",
            "domain_confusion": f"{self.domain_entropy_sum / max(self.n_batches, 1):.4f}",
        }

    def get_confusion_metrics(self):
        return {
            "domain_entropy": self.domain_entropy_sum / max(self.n_batches, 1),
            "feature_alignment": self.feature_alignment_sum / max(self.n_batches, 1),
        }


def confusion_matrix(outputs, masks, num_classes):
    """[num_classes, num_classes] int64 confusion matrix (rows = target, cols = argmax prediction) on the device:
    one HIP kernel (per-pixel argmax + LDS histogram), no intermediate argmax tensor."""
    from . import kernels as K
    from .losses import _padded_nhwc
    if outputs.device.type != "cuda":
        raise RuntimeError("confusion_matrix: logits must live on the GPU (no CPU path in this build)")
    buf, ldc = _padded_nhwc(outputs.detach())
    cm = torch.zeros(num_classes * num_classes, dtype=torch.int64, device=outputs.device)
    tgt = masks.reshape(-1)
    if tgt.dtype != torch.int64:
        tgt = tgt.long()
    K.argmax_confusion(buf, tgt.contiguous(), tgt.numel(), num_classes, ldc, cm)
    return cm.view(num_classes, num_classes)


def segmentation_metrics(outputs, masks, num_classes):
    """{'iou': macro Jaccard over the classes present, 'accuracy', 'iou_class_k': binary Jaccard of class k}.

    Definitions follow torchmetrics' JaccardIndex (multiclass macro: classes absent from both prediction and target
    carry no weight; binary: tp / (tp + fp + fn), 0 when empty), which the reference instantiates at
    ``src/models/train.py:209-222``.  The confusion matrix comes from one HIP kernel; ONE 23x23 transfer to the host.
    """
    cm = confusion_matrix(outputs, masks, num_classes).cpu().double()
    tp = cm.diag()
    denom = cm.sum(0) + cm.sum(1) - tp
    iou_c = torch.where(denom > 0, tp / denom.clamp_min(1), torch.zeros_like(tp))
    present = (cm.sum(0) + cm.sum(1)) > 0
    macro = float((iou_c * present).sum() / present.sum().clamp_min(1))
    acc = float(tp.sum() / cm.sum().clamp_min(1))
    out = {"iou": macro, "accuracy": acc}
    for c in range(num_classes):
        out[f"iou_class_{c}"] = float(iou_c[c])
    return out


class SegmentationMetrics:
    """Mirror of reference ``src/analysis/metrics.py::SegmentationMetrics`` (same constructor, same methods and return
    shapes).  ``predictions`` may be what the reference passes -- an integer class map -- or the ``[N,C,H,W]`` logits
    themselves: then the argmax and the histogram are ONE HIP kernel (``udaseg_argmax_confusion``) and nothing but the
    C x C matrix leaves the device."""

    def __init__(self, num_classes, ignore_index=None):
        self.num_classes = num_classes
        self.ignore_index = ignore_index

    def _fast_hist(self, pred, true):
        """Confusion matrix (numpy int64, rows = target, columns = prediction); targets outside [0, num_classes) and
        ``ignore_index`` are left out (reference :17-29)."""
        k = self.num_classes
        if pred.is_floating_point():
            if pred.dim() != 4 or pred.shape[1] != k:
                raise ValueError(f"logits must be [N,{k},H,W], got {tuple(pred.shape)}")
            hist = confusion_matrix(pred, true, k)
        else:
            if pred.device.type != "cuda":
                raise RuntimeError("SegmentationMetrics: tensors must live on the GPU (no CPU path in this build)")
            t, p = true.reshape(-1).long(), pred.reshape(-1).long()
            m = (t >= 0) & (t < k)
            hist = torch.bincount(k * t[m] + p[m], minlength=k * k).reshape(k, k)
        hist = hist.cpu().numpy().astype("int64")
        if self.ignore_index is not None and 0 <= self.ignore_index < k:
            hist[self.ignore_index, :] = 0
        return hist

    def batch_iou(self, predictions, targets):
        import numpy as np
        hist = self._fast_hist(predictions, targets)
        d = np.diag(hist)
        iu = d / (hist.sum(axis=1) + hist.sum(axis=0) - d + 1e-7)
        return {"mean_iou": np.nanmean(iu), "class_iou": {i: v for i, v in enumerate(iu)}}

    def pixel_accuracy(self, predictions, targets):
        """Reference :47-52 (out-of-range targets stay in the denominator, as upstream)."""
        if predictions.is_floating_point():
            predictions = predictions.argmax(dim=1)
        mask = targets != self.ignore_index if self.ignore_index is not None else torch.ones_like(targets, dtype=torch.bool)
        both = torch.stack([((predictions == targets) & mask).sum(), mask.sum()]).cpu()
        return int(both[0]) / (int(both[1]) + 1e-7)

    def f1_score(self, predictions, targets, class_index=None):
        import numpy as np
        hist = self._fast_hist(predictions, targets)
        tp = np.diag(hist)
        fp, fn = hist.sum(axis=0) - tp, hist.sum(axis=1) - tp
        f1 = 2 * tp / (2 * tp + fp + fn + 1e-7)
        return f1[class_index] if class_index is not None else f1.tolist()
