"""CPU oracle: the reference's segmentation metrics (``src/analysis/metrics.py:5-67``) restated with numpy.

TEST INFRASTRUCTURE ONLY (tests/ import it; nothing in the product package does).

Pinned by ``tests/golden/seg_metrics_ref.npz``: values produced by the REFERENCE's own ``SegmentationMetrics`` on seeded
logits / targets (``oracle/gen_golden.py::gen_seg_metrics``, run in the build container where ``src.analysis.metrics``
imports); ``tests/test_oracle_metrics.py`` requires this restatement to reproduce them exactly.
"""
import numpy as np


def fast_hist(pred, true, num_classes, ignore_index=None):
    """Confusion matrix, rows = target, columns = prediction; targets outside [0, num_classes) (and ``ignore_index``) are
    dropped (reference ``_fast_hist``, :17-29)."""
    pred, true = np.asarray(pred).reshape(-1), np.asarray(true).reshape(-1)
    mask = (true >= 0) & (true < num_classes)
    if ignore_index is not None:
        mask &= true != ignore_index
    return np.bincount(num_classes * true[mask].astype(np.int64) + pred[mask], minlength=num_classes ** 2).reshape(
        num_classes, num_classes)


def batch_iou(hist):
    """(mean IoU, per-class IoU) with the reference's 1e-7 in the denominator (:31-45)."""
    d = np.diag(hist)
    iu = d / (hist.sum(axis=1) + hist.sum(axis=0) - d + 1e-7)
    return float(np.nanmean(iu)), iu


def pixel_accuracy(pred, true, ignore_index=None):
    """:47-52 -- note: unlike the histogram, out-of-range targets still count in the denominator."""
    pred, true = np.asarray(pred), np.asarray(true)
    mask = true != ignore_index if ignore_index is not None else np.ones_like(true, dtype=bool)
    return float(((pred == true) & mask).sum() / (mask.sum() + 1e-7))


def f1_scores(hist):
    """:54-67"""
    tp = np.diag(hist)
    fp, fn = hist.sum(axis=0) - tp, hist.sum(axis=1) - tp
    return 2 * tp / (2 * tp + fp + fn + 1e-7)
