"""HIP loss family (csrc/losses_seg.hip via losses.DiceLoss / WeightedSegmentationLoss / ConsistencyLoss /
FineTuningLoss) against (1) the fixture the REFERENCE's own classes produced (tests/golden/losses_ref.npz, float64 run =
anchor) and (2) the CPU oracle on fresh seeds; then size-independent properties at BASELINE's full 8 x 23 x 512 x 512.

Tolerance: north_star's 1e-3 relative (max-norm) for fp32; values and gradients here land around 1e-6.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses_ref.npz"))
CASES = ("c23", "c5", "c2")
TOL = 1e-3


@pytest.fixture(scope="module")
def L():
    from uda_aerial_semantic_segmentation_research_amd import _lib, losses
    _lib.require_gpu()
    return losses


def rel(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def case_inputs(name):
    from oracle.losses_ref import loss_inputs
    seed, batch, classes, h, w = (int(v) for v in GOLD[f"{name}/shape"])
    return classes, loss_inputs(seed, batch, classes, h, w, dtype=torch.float32)


def hip_fns(L, classes, target, weights):
    return {
        "dice": lambda z: L.DiceLoss()(z, target),
        "dice_smooth": lambda z: L.DiceLoss(smooth=0.1)(z, target),
        "focal": lambda z: L.WeightedSegmentationLoss(classes, weights).focal_loss(z, target),
        "wseg": lambda z: L.WeightedSegmentationLoss(classes, weights)(z, target, 0.7),
        "wseg_sum": lambda z: L.WeightedSegmentationLoss(classes, None, alpha=0.5, gamma=1.5, reduction='sum')(z, target),
        "cons": lambda a, b: L.ConsistencyLoss()(a, b),
        "cons_t2": lambda a, b: L.ConsistencyLoss(2.0)(a, b),
        "fine": lambda a, b, d, s: L.FineTuningLoss()(a, b, d, 10, s, target)['total'],
    }


def inputs_of(key, z1, z2, domain):
    return {"cons": (z1, z2), "cons_t2": (z1, z2), "fine": (z1, z2, domain, z2)}.get(key, (z1,))


@pytest.mark.parametrize("name", CASES)
def test_hip_losses_match_reference_fixture(L, name):
    classes, (z1, z2, target, weights, domain) = case_inputs(name)
    z1, z2, target, weights, domain = (t.cuda() for t in (z1, z2, target, weights, domain))
    worst = 0.0
    for key, fn in hip_fns(L, classes, target, weights).items():
        leaves = [t.clone().requires_grad_(True) for t in inputs_of(key, z1, z2, domain)]
        val = fn(*leaves)
        val.backward()
        ev = rel(val, GOLD[f"{name}/f64/{key}/value"])
        assert ev < 1e-5, (key, "value", ev)
        for i, t in enumerate(leaves):
            eg = rel(t.grad, GOLD[f"{name}/f64/{key}/grad{i}"])
            worst = max(worst, eg)
            assert eg < TOL, (key, i, eg)
    print(f"{name}: worst gradient error vs reference fixture {worst:.2e}")


def test_fine_tuning_dict(L):
    name = "c23"
    classes, (z1, z2, target, weights, domain) = case_inputs(name)
    z1, z2, target, domain = (t.cuda() for t in (z1, z2, target, domain))
    ft = L.FineTuningLoss(0.8, 0.2, 0.3, rampup_length=8)
    keys = list(GOLD[f"{name}/fine_dict/keys"])
    for epoch in (0, 3, 8, 50):
        d = ft(z1, z2, domain, epoch, z2, target.float())
        assert sorted(d) == keys
        assert rel([d[k].item() for k in keys], GOLD[f"{name}/fine_dict/{epoch}"]) < 1e-5
        assert d['total'].requires_grad is False or d['total'].grad_fn is not None
        assert not d['consistency'].requires_grad
    d = ft(z1, z2, domain, 3)
    assert d['supervised'].item() == 0.0
    assert rel([d[k].item() for k in keys], GOLD[f"{name}/fine_dict/unsup"]) < 1e-5


@pytest.mark.parametrize("classes,shape", [(23, (2, 33, 47)), (1, (2, 8, 8)), (32, (1, 16, 16)), (4, (3, 5, 300))])
def test_hip_losses_match_oracle_fresh_seeds(L, classes, shape):
    """Ragged sizes (pixels not a multiple of the block), 1 class, the 32-class maximum, padded and unpadded channel counts."""
    from oracle import losses_ref as O
    b, h, w = shape
    z1, z2, target, weights, domain = O.loss_inputs(100 + classes, b, classes, h, w, dtype=torch.float64)
    if classes == 1:
        target = torch.zeros_like(target)
    ora = {
        "dice": lambda z: O.DiceLossRef()(z, target),
        "wseg": lambda z: O.WeightedSegmentationLossRef(classes, weights, 0.3, 3.0)(z, target, 1.5),
        "cons": lambda a, b_: O.ConsistencyLossRef(0.7)(a, b_),
    }
    tg, wg = target.cuda(), weights.float().cuda()
    hip = {
        "dice": lambda z: L.DiceLoss()(z, tg),
        "wseg": lambda z: L.WeightedSegmentationLoss(classes, wg, 0.3, 3.0)(z, tg, 1.5),
        "cons": lambda a, b_: L.ConsistencyLoss(0.7)(a, b_),
    }
    for key in ora:
        ins = (z1, z2) if key == "cons" else (z1,)
        lo = [t.clone().requires_grad_(True) for t in ins]
        vo = ora[key](*lo)
        vo.backward()
        lh = [t.float().cuda().requires_grad_(True) for t in ins]
        vh = hip[key](*lh)
        (vh * 2.5).backward()                                  # upstream gradient != 1
        assert abs(vh.item() - vo.item()) <= 1e-5 * max(abs(vo.item()), 1e-3), (key, vh.item(), vo.item())
        for a, o in zip(lh, lo):
            if o.grad.abs().max() == 0:
                assert a.grad.abs().max().item() < 1e-5      # exact zero in the oracle (1 class); fp32 rounding here
            else:
                assert rel(a.grad, o.grad * 2.5) < TOL, (key, rel(a.grad, o.grad * 2.5))


def test_one_hot_targets_and_errors(L):
    classes, (z1, _, target, _, _) = case_inputs("c5")
    z, t = z1.cuda(), target.cuda()
    oh = torch.nn.functional.one_hot(t, classes).permute(0, 3, 1, 2).float()
    assert L.DiceLoss()(z, oh).item() == L.DiceLoss()(z, t).item()
    with pytest.raises(RuntimeError):
        L.DiceLoss()(z1, target)                               # CPU tensors: no CPU path in the product
    with pytest.raises(ValueError):
        L.DiceLoss()(z, t[:, :3])
    with pytest.raises(ValueError):
        L.ConsistencyLoss()(z, z[:, :3])
    with pytest.raises(ValueError):
        L.WeightedSegmentationLoss(7)(z, t)


def test_losses_on_unet_output_zero_copy(L):
    """Logits straight from Unet.forward (padded NHWC storage) flow through the new losses and back into the network."""
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    from oracle.losses_ref import DiceLossRef, WeightedSegmentationLossRef
    torch.manual_seed(0)
    model = Unet("resnet18", encoder_weights=None, in_channels=3, classes=23).cuda().train()
    x = torch.randn(2, 3, 64, 64, device="cuda")
    t = torch.randint(0, 23, (2, 64, 64), device="cuda")
    logits = model(x)
    loss = L.WeightedSegmentationLoss(23)(logits, t) + 0.5 * L.ConsistencyLoss()(logits, logits.detach().roll(1, 0))
    loss.backward()
    g = model.segmentation_head[0].weight.grad if hasattr(model, "segmentation_head") else None
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(q).all() for q in grads) and any(q.abs().max() > 0 for q in grads)
    # value check of the zero-copy path against the oracle on the same logits
    lz = logits.detach().double().cpu()
    want = WeightedSegmentationLossRef(23)(lz, t.cpu())
    got = L.WeightedSegmentationLoss(23)(logits.detach(), t)
    assert abs(got.item() - want.item()) < 1e-5 * abs(want.item())
    assert abs(L.DiceLoss()(logits.detach(), t).item() - DiceLossRef()(lz, t.cpu()).item()) < 1e-5


def test_full_size_properties(L):
    """8 x 23 x 512 x 512 (BASELINE configs[1] logits): properties that need no oracle run of that size."""
    g = torch.Generator(device="cuda").manual_seed(3)
    z1 = torch.randn(8, 23, 512, 512, device="cuda", generator=g) * 2
    z2 = z1 + torch.randn(8, 23, 512, 512, device="cuda", generator=g)
    t = torch.randint(0, 23, (8, 512, 512), device="cuda", generator=g)
    # consistency: zero (value and gradient) on identical inputs, symmetric, positive, gradients swap with the arguments
    a = z1.clone().requires_grad_(True)
    same = L.ConsistencyLoss()(a, z1)
    same.backward()
    assert abs(same.item()) < 1e-3 and a.grad.abs().max().item() < 1e-6
    a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    v12 = L.ConsistencyLoss()(a, b)
    v12.backward()
    a2, b2 = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
    v21 = L.ConsistencyLoss()(b2, a2)
    v21.backward()
    assert v12.item() > 0 and abs(v12.item() - v21.item()) <= 1e-6 * v12.item()
    assert rel(a.grad, a2.grad) < 1e-6 and rel(b.grad, b2.grad) < 1e-6
    # softmax gradients sum to zero over classes at every pixel
    assert a.grad.sum(dim=1).abs().max().item() < 1e-6 * max(a.grad.abs().max().item(), 1e-30) * 23 + 1e-9
    # shift invariance: adding a per-pixel constant to all logits changes nothing
    shift = torch.randn(8, 1, 512, 512, device="cuda", generator=g)
    d0 = L.DiceLoss()(z1, t).item()
    assert 0.0 < d0 < 1.0 and abs(L.DiceLoss()(z1 + shift, t).item() - d0) < 1e-5
    # perfect predictions drive Dice to ~0 and the focal term to ~0
    perfect = torch.nn.functional.one_hot(t, 23).permute(0, 3, 1, 2).float() * 60.0
    assert L.DiceLoss()(perfect, t).item() < 1e-5
    ws = L.WeightedSegmentationLoss(23)
    assert ws.focal_loss(perfect, t).item() < 1e-6
    # gradient is linear in the upstream gradient, and Dice + focal gradients add
    c = z1.clone().requires_grad_(True)
    ws(c, t).backward()
    c3 = z1.clone().requires_grad_(True)
    (3.0 * ws(c3, t)).backward()
    assert rel(c3.grad, 3.0 * c.grad) < 1e-6
    cf, cd = z1.clone().requires_grad_(True), z1.clone().requires_grad_(True)
    ws.focal_loss(cf, t).backward()
    L.DiceLoss()(cd, t).backward()
    assert rel(c.grad, cf.grad + cd.grad) < 1e-5
    assert c.grad.sum(dim=1).abs().max().item() < 1e-9


def test_segmentation_metrics_vs_reference_vectors(golden_dir):
    """udaseg_argmax_confusion and metrics.SegmentationMetrics against values produced by the REFERENCE's
    src/analysis/metrics.py::SegmentationMetrics (tests/golden/seg_metrics_ref.npz): the confusion matrix bit for bit
    (void / out-of-range labels dropped), IoU / pixel accuracy / F1 to the last digit, from logits and from class maps."""
    import os
    import numpy as np
    from uda_aerial_semantic_segmentation_research_amd.metrics import SegmentationMetrics, confusion_matrix
    g = np.load(os.path.join(golden_dir, "seg_metrics_ref.npz"))
    logits, target = torch.from_numpy(g["logits"]).cuda(), torch.from_numpy(g["target"]).cuda()
    assert np.array_equal(confusion_matrix(logits, target, 23).cpu().numpy(), g["plain/hist"])
    # the zero-copy path: logits as the padded NHWC view the network hands out
    padded = torch.zeros(2, 24, 40, 24, device="cuda")
    padded[..., :23] = logits.permute(0, 2, 3, 1)
    view = padded.permute(0, 3, 1, 2)[:, :23]
    assert np.array_equal(confusion_matrix(view, target, 23).cpu().numpy(), g["plain/hist"])
    pred = logits.argmax(1)
    for tag, ign in (("plain", None), ("ignore0", 0)):
        m = SegmentationMetrics(23, ignore_index=ign)
        for p in (logits, pred):
            assert np.array_equal(m._fast_hist(p, target), g[f"{tag}/hist"])
            r = m.batch_iou(p, target)
            assert r["mean_iou"] == float(g[f"{tag}/mean_iou"])
            assert np.array_equal(np.array([r["class_iou"][i] for i in range(23)]), g[f"{tag}/class_iou"])
            assert m.pixel_accuracy(p, target) == float(g[f"{tag}/pixel_accuracy"])
            assert np.array_equal(np.array(m.f1_score(p, target)), g[f"{tag}/f1"])
            assert m.f1_score(p, target, class_index=7) == float(g[f"{tag}/f1_class7"])
