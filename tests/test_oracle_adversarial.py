"""Oracle pinning, part 2: the restated discriminator / adversarial loss / metrics / step order against
vectors produced by the REFERENCE's own classes (oracle/gen_golden.py imported src.models.{discriminator,
losses,metrics} in the build container; bit-equality of restatement vs import was asserted there)."""
import os

import numpy as np
import torch

from oracle.adversarial_ref import (AdversarialLossRef, DomainAdaptationMetricsRef, DomainDiscriminatorRef,
                                    adversarial_step, synthetic_batch)
from oracle.unet_ref import UnetRef


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "adversarial_ref.npz"))


def _stats(t):
    f = t.detach().double().flatten()
    return np.array([f.sum().item(), f.abs().sum().item()])



def _stats_close(got, want, rtol=2e-4):
    """[sum, abs-sum] fingerprints: the plain sum cancels, so its tolerance is relative to the abs-sum."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert abs(got[1] - want[1]) <= rtol * abs(want[1]) + 1e-12, (got, want)
    assert abs(got[0] - want[0]) <= rtol * abs(want[1]) + 1e-12, (got, want)


def test_known_answer_losses(golden_dir):
    g = _g(golden_dir)
    assert bool(g["restatement_bit_exact"])
    L = AdversarialLossRef(0.001)
    p_s, p_t = torch.from_numpy(g["ka/p_s"]), torch.from_numpy(g["ka/p_t"])
    # survey-recorded literals from importing the reference
    assert abs(float(g["ka/d_loss"]) - 0.7226598858833313) < 1e-7
    assert abs(float(g["ka/g_loss"]) - 0.00048734844313003123) < 1e-10
    np.testing.assert_allclose(L.discriminator_loss(p_s, p_t).item(), g["ka/d_loss"], rtol=1e-6)
    np.testing.assert_allclose(L.generator_loss(p_t).item(), g["ka/g_loss"], rtol=1e-6)
    # F7: the loss treats probabilities as logits
    want = (torch.nn.functional.softplus(-p_s).mean() + torch.nn.functional.softplus(p_t).mean()) / 2
    np.testing.assert_allclose(L.discriminator_loss(p_s, p_t).item(), want.item(), rtol=1e-6)


def test_discriminator_forward_grads_bn_state(golden_dir):
    g = _g(golden_dir)
    torch.manual_seed(1234)
    D = DomainDiscriminatorRef(3).train()
    assert sum(p.numel() for p in D.parameters()) == 2758849
    for k, v in D.state_dict().items():
        if v.dtype.is_floating_point:
            _stats_close(_stats(v), g["d_init/" + k + "/stats"], rtol=1e-9)
    src, _, tgt = synthetic_batch(2, 64, 64, seed=0)
    L = AdversarialLossRef(0.001)
    ps, pt = D(src), D(tgt)
    assert ps.shape == (2, 1) and float(ps.detach().min()) >= 0 and float(ps.detach().max()) <= 1   # test_system.py:298-301
    dl = L.discriminator_loss(ps, pt)
    assert dl.dim() == 0                                                            # test_system.py:315
    dl.backward()
    np.testing.assert_allclose(ps.detach().numpy(), g["d/p_s"], rtol=1e-5)
    np.testing.assert_allclose(pt.detach().numpy(), g["d/p_t"], rtol=1e-5)
    np.testing.assert_allclose(dl.item(), g["d/d_loss"], rtol=1e-6)
    for k, p in D.named_parameters():
        _stats_close(_stats(p.grad), g["d_grad/" + k + "/stats"])
    np.testing.assert_allclose(L.generator_loss(D(tgt)).item(), g["d/g_loss"], rtol=1e-6)
    for k, v in D.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.numpy(), g["d_bn3/" + k], rtol=1e-5, atol=1e-7)
        if "num_batches" in k:
            assert int(v) == int(g["d_bn3/" + k]) == 3


def test_metrics_strings(golden_dir):
    g = _g(golden_dir)
    torch.manual_seed(1234)
    D = DomainDiscriminatorRef(3).train()
    src, _, tgt = synthetic_batch(2, 64, 64, seed=0)
    ps, pt = D(src).detach(), D(tgt).detach()
    M = DomainAdaptationMetricsRef()
    M.update(ps, pt)
    M.update(pt, ps)
    m = M.get_metrics()
    assert sorted(m.keys()) == list(g["metrics/keys"]) == ["domain_confusion", "source_domain_acc", "target_domain_acc"]
    assert [m[k] for k in sorted(m.keys())] == list(g["metrics/vals"])


def test_full_adversarial_iteration(golden_dir):
    g = _g(golden_dir)
    torch.manual_seed(1234)
    model = UnetRef("resnet18", classes=23).train()
    D = DomainDiscriminatorRef(3).train()
    M = DomainAdaptationMetricsRef()
    L = AdversarialLossRef(0.001)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    dopt = torch.optim.Adam(D.parameters(), lr=opt.param_groups[0]["lr"])
    src, masks, tgt = synthetic_batch(2, 64, 64, seed=0)
    r = adversarial_step(model, D, L, opt, dopt, src, masks, tgt, metrics=M)
    for k in ("seg_loss", "d_loss", "adv_loss", "total"):
        np.testing.assert_allclose(r[k].item(), g["advstep/" + k], rtol=1e-5)
    for k, v in D.state_dict().items():
        if v.dtype.is_floating_point:
            _stats_close(_stats(v), g["advstep/D/" + k + "/stats"], rtol=1e-5)
    sd = model.state_dict()
    for k in ("encoder.conv1.weight", "segmentation_head.0.weight", "encoder.bn1.running_var"):
        _stats_close(_stats(sd[k]), g["advstep/model/" + k + "/stats"], rtol=1e-5)
    # F8: adversarial term adds nothing to the segmenter's gradients
    torch.manual_seed(1234)
    m2 = UnetRef("resnet18", classes=23).train()
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-4)
    from oracle.adversarial_ref import segmentation_step
    segmentation_step(m2, o2, src, masks)
    for (k, a), (_, b) in zip(model.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
