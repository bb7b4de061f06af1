"""GPU parity, network level: the HIP-backed Unet / DomainDiscriminator / losses / Adam against the CPU oracle on the
same seeded inputs and the same weights, plus the committed golden vectors.

Bar (north_star): logits, loss and gradients within 1e-3 relative fp32 -- applied per tensor, norm-wise
(max |diff| / max |ref|).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-3


def rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def check(got, ref, what, rtol=RTOL):
    assert tuple(got.shape) == tuple(ref.shape), (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite"
    e = rel(got, ref)
    assert e <= rtol, f"{what}: rel err {e:.3e} > {rtol:g}"
    return e


@pytest.fixture(scope="module")
def pkg():
    from uda_aerial_semantic_segmentation_research_amd import _lib
    _lib.require_gpu()
    import uda_aerial_semantic_segmentation_research_amd as p
    return p


def _pair(name, classes=23):
    from oracle.unet_ref import UnetRef
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    torch.manual_seed(1234)
    ref = UnetRef(name, classes=classes).train()
    net = Unet(encoder_name=name, encoder_weights=None, in_channels=3, classes=classes)
    net.load_state_dict(ref.state_dict())
    return ref, net.to("cuda").train()


@pytest.mark.parametrize("name", ["resnet18", "resnet34", "resnet50"])
def test_unet_forward_backward_adam_vs_oracle(pkg, name, golden_dir):
    from oracle.adversarial_ref import synthetic_batch
    from uda_aerial_semantic_segmentation_research_amd.losses import CrossEntropyLoss
    from uda_aerial_semantic_segmentation_research_amd.optim import FusedAdam
    ref, net = _pair(name)
    x, y, _ = synthetic_batch(2, 64, 64, seed=0)

    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-4)
    opt_ref.zero_grad()
    logits_ref = ref(x)
    loss_ref = torch.nn.functional.cross_entropy(logits_ref, y)
    loss_ref.backward()

    opt = FusedAdam(net.parameters(), lr=1e-4)
    opt.zero_grad()
    logits = net(x.cuda())
    assert logits.shape == (2, 23, 64, 64)
    loss = CrossEntropyLoss()(logits, y.cuda())
    assert loss.dim() == 0 and loss.grad_fn is not None
    loss.backward()

    check(logits, logits_ref, "logits")
    assert abs(loss.item() - loss_ref.item()) <= RTOL * abs(loss_ref.item())
    worst = ("", 0.0)
    gref = dict(ref.named_parameters())
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        e = check(p.grad, gref[k].grad, f"grad {k}")
        if e > worst[1]:
            worst = (k, e)
    print(f"{name}: worst grad rel err {worst[1]:.2e} at {worst[0]}")
    # BN running statistics after one training forward
    sd, sdr = net.state_dict(), ref.state_dict()
    for k in sdr:
        if "running" in k:
            check(sd[k], sdr[k], k, 1e-4)
        if "num_batches" in k:
            assert int(sd[k]) == int(sdr[k]) == 1
    # golden fixture (made by the oracle in the build container)
    if name in ("resnet18", "resnet50"):
        g = np.load(os.path.join(golden_dir, "unet_oracle.npz"))
        assert abs(loss.item() - float(g[f"{name}/loss"])) <= RTOL * float(g[f"{name}/loss"])
        step = max(1, logits.numel() // 64)
        got = logits.detach().cpu().contiguous().flatten()[::step][:64].numpy()
        want = g[f"{name}/logits/sample"]
        assert np.abs(got - want).max() <= RTOL * np.abs(logits_ref.detach().numpy()).max()
    # one optimizer step: fused flat Adam vs torch.optim.Adam.  A weight moves by ~lr whatever its gradient, so
    # compare the UPDATE, restricted to entries whose gradient is far above rounding noise.
    before = {k: v.detach().clone() for k, v in ref.named_parameters()}
    opt_ref.step()
    opt.step()
    for k, p in net.named_parameters():
        gr = gref[k].grad
        big = gr.abs() > 1e-3 * gr.abs().max()
        upd = (p.detach().cpu() - before[k])[big]
        upd_ref = (gref[k].detach() - before[k])[big]
        assert (upd - upd_ref).abs().max() <= 0.02 * 1e-4 + 1e-9, k
    assert ("flat", 0) in opt.state, "FusedAdam did not take the flat-arena path"


def test_unet_eval_mode_and_state_dict_roundtrip(pkg):
    from oracle.adversarial_ref import synthetic_batch
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    ref, net = _pair("resnet18")
    x, _, _ = synthetic_batch(2, 64, 96, seed=3)
    with torch.no_grad():
        ref.train()(x)                      # move the running stats off their init
    net.load_state_dict(ref.state_dict())
    ref.eval()
    net.eval()
    with torch.no_grad():
        out = net(x.cuda())
        check(out, ref(x), "eval logits")
    assert not out.requires_grad
    # model_io_suite shape (reference src/test_system.py:252-266): save -> new model -> load -> same output
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    net2 = Unet("resnet18", encoder_weights=None, in_channels=3, classes=23).to("cuda").eval()
    net2.load_state_dict(sd)
    with torch.no_grad():
        assert torch.equal(net2(x.cuda()), out)
    with pytest.raises(RuntimeError, match="divisible by 32"):
        net(torch.zeros(1, 3, 40, 64, device="cuda"))


def test_unet_foreign_gradient_layout(pkg):
    """A gradient that does not come from our CE kernel (plain torch ops on the logits) takes the copy path."""
    from oracle.adversarial_ref import synthetic_batch
    ref, net = _pair("resnet18")
    x, _, _ = synthetic_batch(1, 32, 32, seed=5)
    wmap = torch.randn(1, 23, 32, 32)
    (ref(x) * wmap).sum().backward()
    (net(x.cuda()) * wmap.cuda()).sum().backward()
    gref = dict(ref.named_parameters())
    for k, p in net.named_parameters():
        check(p.grad, gref[k].grad, f"grad {k}")


def test_discriminator_vs_reference_vectors(pkg, golden_dir):
    """Forward / loss / gradients / BN state against vectors made by the REFERENCE's own DomainDiscriminator and
    AdversarialLoss (oracle/gen_golden.py), and against the restated oracle tensor by tensor."""
    from oracle.adversarial_ref import AdversarialLossRef, DomainDiscriminatorRef, synthetic_batch
    from uda_aerial_semantic_segmentation_research_amd.discriminator import DomainDiscriminator
    from uda_aerial_semantic_segmentation_research_amd.losses import AdversarialLoss
    g = np.load(os.path.join(golden_dir, "adversarial_ref.npz"))
    torch.manual_seed(1234)
    ref = DomainDiscriminatorRef(3).train()
    D = DomainDiscriminator(input_channels=3)
    D.load_state_dict(ref.state_dict())
    D = D.to("cuda").train()
    src, _, tgt = synthetic_batch(2, 64, 64, seed=0)
    L, Lr = AdversarialLoss(0.001), AdversarialLossRef(0.001)

    ps_r, pt_r = ref(src), ref(tgt)
    dl_r = Lr.discriminator_loss(ps_r, pt_r)
    dl_r.backward()
    ps, pt = D(src.cuda()), D(tgt.cuda())
    assert ps.shape == (2, 1) and float(ps.min()) >= 0 and float(ps.max()) <= 1       # test_system.py:298-301
    dl = L.discriminator_loss(ps, pt)
    assert dl.dim() == 0                                                                # test_system.py:315
    dl.backward()
    np.testing.assert_allclose(ps.detach().cpu().numpy(), g["d/p_s"], rtol=RTOL)
    np.testing.assert_allclose(pt.detach().cpu().numpy(), g["d/p_t"], rtol=RTOL)
    assert abs(dl.item() - float(g["d/d_loss"])) <= 1e-5
    gr = dict(ref.named_parameters())
    for k, p in D.named_parameters():
        if k in ("features.2.bias", "features.5.bias", "features.8.bias"):
            # a bias in front of BatchNorm has an exactly-zero gradient; both sides hold rounding noise only
            assert p.grad.abs().max().item() <= 1e-6 and gr[k].grad.abs().max().item() <= 1e-6
            continue
        check(p.grad, gr[k].grad, f"D grad {k}")
        s = np.array([p.grad.double().sum().item(), p.grad.double().abs().sum().item()])
        w = g["d_grad/" + k + "/stats"]
        assert abs(s[1] - w[1]) <= RTOL * w[1] and abs(s[0] - w[0]) <= RTOL * w[1], k
    # third train-mode forward (adversarial_trainer.py:108): generator loss + BN running state after 3 passes
    gl = L.generator_loss(D(tgt.cuda()))
    assert abs(gl.item() - float(g["d/g_loss"])) <= 1e-7
    for k, v in D.state_dict().items():
        if "running" in k:
            np.testing.assert_allclose(v.cpu().numpy(), g["d_bn3/" + k], rtol=1e-4, atol=1e-6)
        if "num_batches" in k:
            assert int(v) == 3


def test_full_adversarial_iteration_vs_oracle_and_golden(pkg, golden_dir):
    """One iteration in the reference's order (adversarial_trainer.py:85-114) through the build's trainer."""
    from oracle.adversarial_ref import (AdversarialLossRef, DomainAdaptationMetricsRef, DomainDiscriminatorRef,
                                        adversarial_step, synthetic_batch)
    from uda_aerial_semantic_segmentation_research_amd.adversarial_trainer import AdversarialTrainer
    from uda_aerial_semantic_segmentation_research_amd.optim import FusedAdam
    g = np.load(os.path.join(golden_dir, "adversarial_ref.npz"))
    ref, net = _pair("resnet18")
    torch.manual_seed(1234)
    from oracle.unet_ref import UnetRef
    _ = UnetRef("resnet18", classes=23)                 # consume the same RNG draws as gen_golden before D's init
    Dr = DomainDiscriminatorRef(3).train()
    src, masks, tgt = synthetic_batch(2, 64, 64, seed=0)
    opt_r = torch.optim.Adam(ref.parameters(), lr=1e-4)
    dopt_r = torch.optim.Adam(Dr.parameters(), lr=1e-4)
    Mr = DomainAdaptationMetricsRef()

    tr = AdversarialTrainer(net, "cuda", lambda_adv=0.001)
    tr.discriminator.load_state_dict(Dr.state_dict())
    opt = FusedAdam(net.parameters(), lr=1e-4)
    out = tr.train_epoch([(src, masks)], [tgt], opt, epoch=1)
    r = adversarial_step(ref, Dr, AdversarialLossRef(0.001), opt_r, dopt_r, src, masks, tgt, metrics=Mr)

    avg_loss, dm = out
    assert abs(avg_loss - r["total"].item()) <= RTOL * abs(r["total"].item())
    assert abs(avg_loss - float(g["advstep/total"])) <= RTOL * float(g["advstep/total"])
    assert dm == Mr.get_metrics()
    assert set(dm) == {"source_domain_acc", "target_domain_acc", "domain_confusion"}     # test_system.py:445-449
    assert list(g["advstep/metrics"]) == [dm[k] for k in sorted(dm)]
    last = tr.last_losses
    for k in ("seg_loss", "d_loss", "adv_loss"):
        assert abs(last[k] - r[k].item()) <= RTOL * abs(r[k].item()) + 1e-9, k
        assert abs(last[k] - float(g["advstep/" + k])) <= RTOL * abs(float(g["advstep/" + k])) + 1e-9, k
    # D's BN running stats saw three forwards; weights moved by one Adam step each
    sdr = Dr.state_dict()
    for k, v in tr.discriminator.state_dict().items():
        if "running" in k:
            check(v, sdr[k], "D " + k, 1e-4)
    assert tr.discriminator_optimizer.param_groups[0]["lr"] == 1e-4                      # lr copied lazily (:55-59)


def test_domain_adaptation_suite_shapes(pkg):
    """Mirror of the reference's domain_adaptation_suite (src/test_system.py:289-329), tensors on the device."""
    from uda_aerial_semantic_segmentation_research_amd.discriminator import DomainDiscriminator
    from uda_aerial_semantic_segmentation_research_amd.losses import AdversarialLoss
    D = DomainDiscriminator(input_channels=3).to("cuda")
    out = D(torch.randn(4, 3, 256, 256, device="cuda"))
    assert out.shape == (4, 1) and torch.all((out >= 0) & (out <= 1))
    L = AdversarialLoss(lambda_adv=0.001)
    d = L.discriminator_loss(torch.rand(4, 1, device="cuda"), torch.rand(4, 1, device="cuda"))
    gl = L.generator_loss(torch.rand(4, 1, device="cuda"))
    assert isinstance(d, torch.Tensor) and d.dim() == 0 and isinstance(gl, torch.Tensor) and gl.dim() == 0
    with pytest.raises(RuntimeError, match="no CPU path"):
        L.generator_loss(torch.rand(4, 1))
