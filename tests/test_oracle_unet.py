"""Oracle pinning, part 1: UnetRef's structure against the reference's own traced-graph fixture,
and the oracle's numeric regression anchors (tests/golden/unet_oracle.npz)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.unet_ref import UnetRef, conv_flops_fwd
from oracle.adversarial_ref import synthetic_batch



def _stats_close(got, want, rtol=2e-4):
    """[sum, abs-sum] fingerprints: the plain sum cancels, so its tolerance is relative to the abs-sum."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert abs(got[1] - want[1]) <= rtol * abs(want[1]) + 1e-12, (got, want)
    assert abs(got[0] - want[0]) <= rtol * abs(want[1]) + 1e-12, (got, want)


def test_r50_matches_reference_trace(golden_dir):
    """Every aten op, output shape and conv/BN/pool hyper-parameter of the model the reference's author
    ran (test_logs/*/events.out.tfevents.*, via tensorboard_logger.py:79-83) is reproduced in order."""
    ref = json.load(open(os.path.join(golden_dir, "unet_r50_trace.json")))["ops"]
    m = UnetRef("resnet50", encoder_weights=None, in_channels=3, classes=23)
    ops = m.trace(torch.zeros(1, 3, 256, 256))
    assert len(ops) == len(ref) == 212
    for a, b in zip(ops, ref):
        assert a["op"] == b["op"] and a["out"] == b["out"], (a, b)
        if a["op"] == "_convolution":
            for k in ("in", "bias", "stride", "padding", "dilation", "groups"):
                assert a[k] == b[k], (k, a, b)
        if a["op"] == "batch_norm":
            assert a["momentum"] == b["momentum"] and a["eps"] == b["eps"]
        if a["op"] == "max_pool2d":
            for k in ("kernel", "stride", "padding", "dilation", "ceil_mode"):
                assert a[k] == b[k]
        if a["op"] == "cat":
            assert a["in_shapes"] == b["in_shapes"] and a["dim"] == b["dim"]  # order [upsampled, skip]
    from collections import Counter
    c = Counter(o["op"] for o in ops)
    assert (c["_convolution"], c["batch_norm"], c["relu_"], c["add_"], c["upsample_nearest2d"], c["cat"]) == \
        (64, 63, 59, 16, 5, 4)


@pytest.mark.parametrize("name,nparams", [("resnet18", 14331399), ("resnet34", 24439559), ("resnet50", 32524295)])
def test_param_counts(name, nparams):
    m = UnetRef(name, classes=23)
    assert sum(p.numel() for p in m.parameters()) == nparams


def test_state_dict_keys_smp_schema():
    sd = UnetRef("resnet18").state_dict()
    for k in ("encoder.conv1.weight", "encoder.bn1.running_var", "encoder.layer2.0.downsample.0.weight",
              "encoder.layer2.0.downsample.1.num_batches_tracked", "decoder.blocks.0.conv1.0.weight",
              "decoder.blocks.4.conv2.1.bias", "segmentation_head.0.weight", "segmentation_head.0.bias"):
        assert k in sd
    assert "encoder.layer1.0.downsample.0.weight" not in sd          # r18: none in layer1
    assert "encoder.layer1.0.downsample.0.weight" in UnetRef("resnet50").state_dict()


def test_flop_model_matches_survey():
    assert abs(conv_flops_fwd(UnetRef("resnet18"), 8, 512, 512) / 1e9 - 358.76) < 0.01
    assert abs(conv_flops_fwd(UnetRef("resnet50"), 8, 768, 768) / 1e9 - 1560.99) < 0.01


def _close(got, want, rtol=2e-4, atol=1e-6):
    np.testing.assert_allclose(np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", ["resnet18", "resnet50"])
def test_oracle_regression_vectors(golden_dir, name):
    """The oracle reproduces its committed vectors (same seeds) -> the fixture travels to the GPU box."""
    g = np.load(os.path.join(golden_dir, "unet_oracle.npz"))
    torch.manual_seed(1234)
    model = UnetRef(name, classes=23).train()
    x, y, _ = synthetic_batch(2, 64, 64, seed=0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    logits = model(x)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    _close(loss.item(), g[f"{name}/loss"], rtol=1e-5)
    step = max(1, logits.numel() // 64)
    _close(logits.detach().flatten()[::step][:64].numpy(), g[f"{name}/logits/sample"], rtol=1e-3, atol=1e-5)
    gw = dict(model.named_parameters())["segmentation_head.0.weight"].grad
    _stats_close([gw.double().sum().item(), gw.double().abs().sum().item()],
                 g[f"{name}/grad/segmentation_head.0.weight/stats"])
    opt.step()
    rv = model.state_dict()["encoder.bn1.running_var"]
    _stats_close([rv.double().sum().item(), rv.double().abs().sum().item()],
                 g[f"{name}/after_adam/encoder.bn1.running_var/stats"], rtol=1e-5)


def test_upsample_mode_switch():
    """nearest is the reference-parity default (trace); bilinear is north_star's named alternate."""
    a = UnetRef("resnet18", upsample="nearest").eval()
    b = UnetRef("resnet18", upsample="bilinear").eval()
    b.load_state_dict(a.state_dict())
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        assert a(x).shape == b(x).shape == (1, 23, 64, 64)
        assert not torch.allclose(a(x), b(x))


def test_baseline_config0_cpu_plumbing():
    """BASELINE.json configs[0]: the reference's own CPU-runnable case (tiny encoder-decoder, 4 x 256 x 256 synthetic source
    batch) through the oracle's train step: finite, and two Adam steps at lr 1e-3 lower the loss on the same batch."""
    from oracle.adversarial_ref import segmentation_step, synthetic_batch
    torch.manual_seed(1234)
    torch.set_num_threads(8)
    model = UnetRef("resnet18", classes=23).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x, y, _ = synthetic_batch(4, 256, 256, seed=0)
    losses = [float(segmentation_step(model, opt, x, y)[0]) for _ in range(3)]
    assert all(np.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
