"""Host-side logic of the trainers (no GPU): early stopping as the reference defines it (src/models/train.py:79-195),
target-loader wrap-around and mask squeezing (src/models/adversarial_trainer.py:69-82), bench/roofline constants."""
import pytest
import torch

from uda_aerial_semantic_segmentation_research_amd.adversarial_trainer import _drop_channel_axis, _wrap_around
from uda_aerial_semantic_segmentation_research_amd.train import EarlyStopping


class _Log:
    def __init__(self):
        self.rows = []

    def log_scalar(self, tag, value, step):
        self.rows.append((tag, value, step))


def test_early_stopping_min_epochs_gate_patience_and_weights():
    es = EarlyStopping(patience=2, min_delta=0.0, mode="min", min_epochs=3, metrics_to_track=["loss", "iou"],
                       weights={"loss": 1.0, "iou": -1.0})
    log = _Log()
    # epochs below min_epochs never stop and never set a best score, but are logged and recorded
    for e, (l, i) in enumerate([(5.0, 0.1), (4.0, 0.2), (3.0, 0.3)]):
        assert es(e, {"loss": l, "iou": i, "accuracy": 0.5}, log) is False
    assert es.best_score is None and es.metric_history == {"loss": [5.0, 4.0, 3.0], "iou": [0.1, 0.2, 0.3]}
    assert [r[0] for r in log.rows[:2]] == ["early_stopping/score", "early_stopping/counter"]
    assert log.rows[0][1] == pytest.approx(4.9)                              # 1.0*5.0 - 1.0*0.1: untracked metrics ignored
    assert es(3, {"loss": 2.0, "iou": 0.4}) is False and es.best_score == pytest.approx(1.6)   # first score after the gate
    assert es(4, {"loss": 1.5, "iou": 0.5}) is False and es.counter == 0     # better
    best = es.get_best_metrics()
    assert best == {"loss": 1.5, "iou": 0.5}
    assert es(5, {"loss": 1.6, "iou": 0.5}) is False and es.counter == 1     # worse once
    assert es(6, {"loss": 1.7, "iou": 0.5}) is True and es.early_stop        # worse twice = patience
    assert es.get_best_metrics() == best
    rates = es.get_improvement_rate()
    assert rates["loss"] == pytest.approx((1.7 - 5.0) / 7)


def test_early_stopping_max_mode_and_min_delta():
    es = EarlyStopping(patience=1, min_delta=0.1, mode="max", min_epochs=0, metrics_to_track=["iou"], weights={"iou": 1.0})
    assert es(0, {"iou": 0.5}) is False
    assert es(1, {"iou": 0.55}) is True                                      # +0.05 < min_delta: not an improvement
    es = EarlyStopping(patience=1, min_delta=0.1, mode="max", min_epochs=0, metrics_to_track=["iou"], weights={"iou": 1.0})
    assert es(0, {"iou": 0.5}) is False and es(1, {"iou": 0.7}) is False and es.best_score == pytest.approx(0.7)


def test_target_loader_wraps_around_and_masks_lose_their_channel_axis():
    it = _wrap_around([1, 2, 3])
    assert [next(it) for _ in range(7)] == [1, 2, 3, 1, 2, 3, 1]
    with pytest.raises(ValueError):
        next(_wrap_around([]))
    m = torch.zeros(4, 1, 8, 8, dtype=torch.long)
    assert _drop_channel_axis(m).shape == (4, 8, 8)
    assert _drop_channel_axis(m[:, 0]).shape == (4, 8, 8)
    assert _drop_channel_axis(torch.zeros(4, 2, 8, 8)).shape == (4, 2, 8, 8)


def test_bench_constants_match_survey():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.FP32_MFMA_PEAK_TFLOPS == 157.3 and bench.BF16_MFMA_PEAK_TFLOPS == 2500.0
    # 133.30 GFLOP of convolution work per source image (r18-Unet at 512x512, fwd + dgrad + wgrad)
    from oracle.unet_ref import UnetRef, conv_flops_fwd
    fwd = conv_flops_fwd(UnetRef("resnet18", classes=23), 1, 512, 512)
    assert abs(fwd / 1e9 - 44.85) < 0.05
    assert abs(bench.CONV_GFLOP_PER_IMAGE[("segmentation", "resnet18", 512)] - 133.30) < 1e-9
    # fwd + dgrad + wgrad = 3 x fwd minus the stem's data gradient (the image needs none): SURVEY B.4
    stem = 2 * 64 * 256 * 256 * 3 * 49 / 1e9
    assert abs(3 * fwd / 1e9 - stem - 133.30) < 0.05
    # the adversarial iteration adds three discriminator passes (fwd + bwd, no dgrad for D's first conv) over 8 + 8 images
    assert abs(bench.CONV_GFLOP_PER_IMAGE[("adversarial", "resnet18", 512)] - 251.68) < 1e-9
    # percentile helper used for the per-step HIP-event report
    assert bench.percentile([5.0, 1.0, 3.0], 0.5) == 3.0 and bench.percentile([1.0], 0.9) == 1.0
