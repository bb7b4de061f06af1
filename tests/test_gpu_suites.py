"""The reference's own suites (src/test_system.py), re-expressed over synthetic loaders on the HIP path, plus checks at
BASELINE.json's full size (8 x 3 x 512 x 512) through properties that do not need an oracle run of that size.

Reference suites mirrored (its dispatcher never calls most of them; the suite FUNCTIONS are the contract):
  model_creation_suite :87-101, training_suite :201-249, model_io_suite :252-266, domain_adaptation_suite :289-329 (in
  test_gpu_model.py), adversarial_training_suite :402-458.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    from uda_aerial_semantic_segmentation_research_amd import _lib
    _lib.require_gpu()
    return True


def _loader(n_batches, bs, hw, classes=23, seed=0, with_masks=True):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_batches):
        x = torch.randn(bs, 3, hw, hw, generator=g)
        if with_masks:
            out.append((x, torch.randint(0, classes, (bs, hw, hw), generator=g)))
        else:
            out.append(x)
    return out


def test_model_creation_suite(pkg):
    """src/test_system.py:87-101 with the reconstructed Config (resnet50, 23 classes, 3 channels, 256x256)."""
    from uda_aerial_semantic_segmentation_research_amd.config import Config
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet, create_model
    model = Unet(encoder_name=Config.ENCODER_NAME, encoder_weights=Config.ENCODER_WEIGHTS, in_channels=Config.IN_CHANNELS,
                 classes=Config.NUM_CLASSES)
    assert sum(p.numel() for p in model.parameters()) == 32524295
    model = model.to("cuda").eval()
    with torch.no_grad():
        out = model(torch.randn(1, 3, *Config.IMAGE_SIZE, device="cuda"))
    assert out.shape == (1, 23, 256, 256) and torch.isfinite(out).all()     # the traced output shape
    assert isinstance(create_model("resnet18", None, 3, 23), Unet)


def test_training_suite(pkg, tmp_path):
    """training_suite: trainer.train(...) for 2 epochs runs, logs the early-stopping scalars, loss goes down."""
    from uda_aerial_semantic_segmentation_research_amd.config import Config
    from uda_aerial_semantic_segmentation_research_amd.train import SegmentationTrainer
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    Config.CHECKPOINTS_DIR = str(tmp_path / "ckpt")
    torch.manual_seed(0)
    model = Unet("resnet18", encoder_weights=None, in_channels=3, classes=23)
    trainer = SegmentationTrainer(model, torch.device("cuda"))
    train_dl, val_dl = _loader(6, 4, 64), _loader(2, 4, 64, seed=1)
    first = trainer.validate(val_dl)
    assert set(first) == {"loss", "iou", "accuracy"}
    trainer.train(train_dl, val_dl, epochs=2, learning_rate=1e-3, patience=7)
    tags = trainer.logger.scalars
    assert "early_stopping/score" in tags and "early_stopping/counter" in tags          # test_system.py:227-242
    assert len(tags["early_stopping/score"]) == 2
    m = trainer.calculate_metrics(model(train_dl[0][0].cuda()).detach(), train_dl[0][1].cuda())
    assert {"iou", "accuracy", "iou_class_0", "iou_class_22"} <= set(m) and 0.0 <= m["iou"] <= 1.0
    # same batch repeatedly: the step must actually learn
    opt_losses = []
    from uda_aerial_semantic_segmentation_research_amd.optim import FusedAdam
    opt = FusedAdam(model.parameters(), lr=1e-3)
    model.train()
    x, y = train_dl[0][0].cuda(), train_dl[0][1].cuda()
    for _ in range(12):
        loss, _ = trainer.train_step(x, y, opt)
        opt_losses.append(loss.item())
    assert all(b < a for a, b in zip(opt_losses, opt_losses[1:])) and opt_losses[-1] < 0.95 * opt_losses[0], opt_losses


def test_model_io_suite(pkg, tmp_path):
    """model_io_suite: save state_dict -> new model -> load -> identical prediction."""
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    m1 = Unet("resnet18", encoder_weights=None, in_channels=3, classes=23).to("cuda").eval()
    path = tmp_path / "model.pth"
    torch.save({"model_state_dict": m1.state_dict()}, path)
    m2 = Unet("resnet18", encoder_weights=None, in_channels=3, classes=23).to("cuda").eval()
    m2.load_state_dict(torch.load(path)["model_state_dict"])
    x = torch.randn(2, 3, 64, 64, device="cuda")
    with torch.no_grad():
        assert torch.equal(m1(x), m2(x))


def test_adversarial_training_suite(pkg):
    """adversarial_training_suite: AdversarialTrainer.train(...) for 2 epochs; domain metric keys present."""
    from uda_aerial_semantic_segmentation_research_amd.adversarial_trainer import AdversarialTrainer
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    torch.manual_seed(0)
    model = Unet("resnet18", encoder_weights=None, in_channels=3, classes=23)
    tr = AdversarialTrainer(model, torch.device("cuda"), lambda_adv=0.001)
    src = [(x, y.unsqueeze(1)) for x, y in _loader(3, 2, 64)]           # [B,1,H,W] masks get squeezed (:80-82)
    tgt = _loader(2, 2, 64, seed=5, with_masks=False)                    # shorter target loader is cycled (:69-73)
    val = _loader(1, 2, 64, seed=7)
    tr.train(src, tgt, val, epochs=2, learning_rate=1e-4, patience=3)
    dm = tr.domain_metrics.get_metrics()
    assert set(dm) == {"source_domain_acc", "target_domain_acc", "domain_confusion"}    # test_system.py:445-449
    assert tr.discriminator_optimizer is not None
    vloss, vm = tr.validate(val)
    assert vloss > 0 and set(vm) == {"iou", "accuracy"}
    for p in tr.discriminator.parameters():
        assert torch.isfinite(p).all()


def test_full_size_step_properties(pkg):
    """BASELINE cfg 2 size (8 x 3 x 512 x 512, r18).  Against the CPU oracle at FULL size: logits, loss, every ReLU output,
    EVERY parameter gradient (teacher-forced masks, with the overridden bits counted and bounded) and the head / last
    decoder block gradients against the oracle's natural backward (tests/_parity.py::grads_vs_oracle).  Then the
    size-independent properties: a repeated step is bitwise reproducible in its forward and reproducible to the fp32
    atomics' ordering noise in its gradients, gradients are linear in the upstream gradient, per-pixel gradient rows of
    the CE sum to zero, BN output statistics are (beta, gamma^2)."""
    from _parity import grads_vs_oracle
    from oracle.adversarial_ref import synthetic_batch
    from oracle.unet_ref import UnetRef
    from uda_aerial_semantic_segmentation_research_amd.losses import CrossEntropyLoss
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    torch.manual_seed(1234)
    ref = UnetRef("resnet18", classes=23).train()
    net = Unet("resnet18", encoder_weights=None, in_channels=3, classes=23)
    net.load_state_dict(ref.state_dict())
    net = net.to("cuda").train()
    x, y, _ = synthetic_batch(8, 512, 512, seed=0)
    xd, yd = x.cuda(), y.cuda()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    with torch.no_grad():
        logits_ref = ref(x)
        loss_ref = torch.nn.functional.cross_entropy(logits_ref, y)
    ref.load_state_dict(state)
    crit = CrossEntropyLoss()
    net.debug_keep_tape = True
    logits = net(xd)
    loss = crit(logits, yd)
    err = (logits.detach().cpu() - logits_ref).abs().max() / logits_ref.abs().max()
    assert err < 1e-3, f"full-size logits rel err {err:.3e}"
    assert abs(loss.item() - loss_ref.item()) < 1e-4 * loss_ref.item()
    loss.backward()
    g1 = net._grad_arena.clone()
    assert torch.isfinite(g1).all()
    grads_vs_oracle(net, ref, x, lambda o: torch.nn.functional.cross_entropy(o, y), "cfg2 r18 8x512x512")
    net.debug_keep_tape = False
    net._last_tape = None
    # the same step again: forward bitwise reproducible; gradients differ by the order of the split-K fp32 atomics only
    net.zero_grad()
    logits2 = net(xd)
    assert torch.equal(logits2.detach(), logits.detach())
    crit(logits2, yd).backward()
    spread = ((net._grad_arena - g1).abs().max() / g1.abs().max()).item()
    # linear in the upstream gradient (x3 is not a power of two: every product re-rounds)
    net.zero_grad()
    (3.0 * crit(net(xd), yd)).backward()
    g3 = net._grad_arena
    rel = ((g3 - 3.0 * g1).abs().max() / g1.abs().max()).item()
    print(f"cfg2 gradients: run-to-run spread {spread:.3e} (fp32 atomics order), linearity under x3 {rel:.3e}")
    assert spread < 2e-6, f"run-to-run gradient spread {spread:.3e}"      # observed 1.4e-7
    assert rel < 2e-5, f"gradient linearity {rel:.3e}"                    # observed 2.3e-6 (round 1 allowed 2e-3)
    lg = logits.detach().requires_grad_(True)
    crit(lg, yd).backward()
    assert lg.grad.sum(dim=1).abs().max().item() < 1e-9
    # BN+ReLU output of the stem in training mode: pre-activation statistics are (beta, gamma^2) = (0, 1) at init
    net.debug_keep_tape = True
    net(xd)
    _, _, (r_stem, f1, _, _), _, _ = net._last_tape
    y_stem, (mean, rstd) = r_stem[4], r_stem[6]
    pre = (y_stem - mean) * rstd
    assert pre.mean(dim=(0, 1, 2)).abs().max().item() < 1e-4
    assert (pre.var(dim=(0, 1, 2), unbiased=False) - 1).abs().max().item() < 1e-3


def test_segmentation_metrics_definitions(pkg):
    """calculate_metrics keys and values against the torchmetrics definitions restated with plain torch ops."""
    from uda_aerial_semantic_segmentation_research_amd.metrics import segmentation_metrics
    g = torch.Generator().manual_seed(3)
    k = 23
    out = torch.randn(2, k, 32, 32, generator=g)
    out[:, 7] += 1.5                                              # skew the predictions so some classes never appear
    masks = torch.randint(0, 12, (2, 32, 32), generator=g)        # classes 12..22 absent from the target
    m = segmentation_metrics(out.cuda(), masks.cuda(), k)
    pred = out.argmax(1)
    ious, present = [], []
    for c in range(k):
        tp = ((pred == c) & (masks == c)).sum().item()
        den = ((pred == c) | (masks == c)).sum().item()
        ious.append(tp / den if den else 0.0)
        present.append(den > 0)
        assert abs(m[f"iou_class_{c}"] - ious[-1]) < 1e-12
    macro = sum(i for i, p in zip(ious, present) if p) / max(sum(present), 1)
    assert abs(m["iou"] - macro) < 1e-12
    assert abs(m["accuracy"] - (pred == masks).float().mean().item()) < 1e-7
