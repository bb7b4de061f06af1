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

from _parity import RTOL, check, grads_vs_oracle, pair as _pair, rel  # noqa: F401  (tests/ is on sys.path via conftest)


@pytest.fixture(scope="module")
def pkg():
    from uda_aerial_semantic_segmentation_research_amd import _lib
    _lib.require_gpu()
    import uda_aerial_semantic_segmentation_research_amd as p
    return p


@pytest.mark.parametrize("name", ["resnet18", "resnet50"])
def test_unet_train_forward_vs_golden_fixture(pkg, name, golden_dir):
    """Train-mode forward + CE at the committed fixture's size (tests/golden/unet_oracle.npz, made by the oracle in the
    build container): loss, logits sample, per-stage encoder features."""
    from oracle.adversarial_ref import synthetic_batch
    from uda_aerial_semantic_segmentation_research_amd.losses import CrossEntropyLoss
    g = np.load(os.path.join(golden_dir, "unet_oracle.npz"))
    _, net = _pair(name)
    x, y, _ = synthetic_batch(2, 64, 64, seed=0)
    with torch.no_grad():
        logits = net(x.cuda())
        loss = CrossEntropyLoss()(logits, y.cuda())
    assert abs(loss.item() - float(g[f"{name}/loss"])) <= 1e-4 * float(g[f"{name}/loss"])
    step = max(1, logits.numel() // 64)
    got = logits.detach().cpu().contiguous().flatten()[::step][:64].numpy()
    want = g[f"{name}/logits/sample"]
    assert np.abs(got - want).max() <= RTOL * np.abs(want).max()
    st = np.array([logits.double().sum().item(), logits.double().abs().sum().item()])
    ws = g[f"{name}/logits/stats"]
    assert abs(st[1] - ws[1]) <= RTOL * ws[1] and abs(st[0] - ws[0]) <= RTOL * ws[1]


# r50 is checked at 128x128: at 64x64 its layer4 BatchNorm sees 8 values per channel, an ill-conditioned regime in which
# a change of fp32 summation ORDER alone moves layer4 gradients by ~1e-3
# ("resnet18", 256, 4) is BASELINE.json configs[0]: the reference's own CPU-runnable training case (4 x 3 x 256 x 256)
@pytest.mark.parametrize("name,size,batch", [("resnet18", 64, 2), ("resnet34", 64, 2), ("resnet50", 128, 2),
                                             ("resnet18", 256, 4)])
def test_unet_forward_backward_adam_vs_oracle(pkg, name, size, batch):
    """One train step (reference src/models/train.py:340-344): logits, loss, EVERY gradient, BN running statistics and the
    Adam update against the CPU oracle."""
    from oracle.adversarial_ref import synthetic_batch
    from uda_aerial_semantic_segmentation_research_amd.losses import CrossEntropyLoss
    from uda_aerial_semantic_segmentation_research_amd.optim import FusedAdam
    ref, net = _pair(name)
    x, y, _ = synthetic_batch(batch, size, size, seed=0)

    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-4)
    opt_ref.zero_grad()
    logits_ref = ref(x)
    loss_ref = torch.nn.functional.cross_entropy(logits_ref, y)
    loss_ref.backward()

    opt = FusedAdam(net.parameters(), lr=1e-4)
    opt.zero_grad()
    net.debug_keep_tape = True
    logits = net(x.cuda())
    assert logits.shape == (batch, 23, size, size)
    loss = CrossEntropyLoss()(logits, y.cuda())
    assert loss.dim() == 0 and loss.grad_fn is not None
    loss.backward()

    check(logits, logits_ref, "logits")
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    grads_vs_oracle(net, ref, x, lambda o: torch.nn.functional.cross_entropy(o, y), f"{name} {batch}x{size}x{size}")
    gref = dict(ref.named_parameters())
    # BN running statistics after one training forward
    sd, sdr = net.state_dict(), ref.state_dict()
    for k in sdr:
        if "running" in k:
            check(sd[k], sdr[k], k, 1e-4)
        if "num_batches" in k:
            assert int(sd[k]) == int(sdr[k]) == 1
    # one optimizer step: fused flat Adam vs torch.optim.Adam.  Adam moves every weight by ~lr whatever the size of
    # its gradient, so compare the UPDATE on entries whose gradient agrees in sign and is far above the noise.
    before = {k: v.detach().clone() for k, v in ref.named_parameters()}
    gpu_grads = {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()}
    opt_ref.step()
    opt.step()
    for k, p in net.named_parameters():
        gr, gg = gref[k].grad, gpu_grads[k]
        sel = (gr.abs() > 0.05 * gr.abs().max()) & ((gr - gg).abs() <= 1e-3 * gr.abs())
        if not sel.any():
            continue
        upd = (p.detach().cpu() - before[k])[sel]
        upd_ref = (gref[k].detach() - before[k])[sel]
        assert (upd - upd_ref).abs().max() <= 0.01 * 1e-4, k
    assert opt.flat_launches == 1, "FusedAdam did not take the flat-arena path"


def _harness(block):
    from uda_aerial_semantic_segmentation_research_amd.engine import ArenaModule

    class H(ArenaModule):
        def __init__(self):
            super().__init__()
            self.block = block
            self.build_arena()
    return H()


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def _nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2)


@pytest.mark.parametrize("kind,cin,planes,stride", [("basic", 64, 64, 1), ("basic", 64, 128, 2), ("bottleneck", 64, 64, 1),
                                                     ("bottleneck", 256, 128, 2), ("bottleneck", 256, 64, 1)])
def test_residual_block_plan(pkg, kind, cin, planes, stride):
    """One residual block, forward + backward plan (identity / downsample wiring, gradient accumulation into an
    already-populated slot) against torch autograd on the oracle's block with the same weights: 2-3 layers deep, so the
    plain 1e-3 bar applies with a wide margin."""
    from oracle import unet_ref as R
    from uda_aerial_semantic_segmentation_research_amd import unet as U
    from uda_aerial_semantic_segmentation_research_amd.engine import GradSlots, Plan
    torch.manual_seed(7)
    exp = 1 if kind == "basic" else 4
    need_ds = stride != 1 or cin != planes * exp
    if kind == "basic":
        rb = R.BasicBlock(cin, planes, stride, torch.nn.Sequential(torch.nn.Conv2d(cin, planes, 1, stride, bias=False),
                                                                   torch.nn.BatchNorm2d(planes)) if need_ds else None)
        mb = U.BasicBlock(cin, planes, stride, need_ds)
    else:
        rb = R.Bottleneck(cin, planes, stride, torch.nn.Sequential(torch.nn.Conv2d(cin, planes * 4, 1, stride, bias=False),
                                                                   torch.nn.BatchNorm2d(planes * 4)) if need_ds else None)
        mb = U.Bottleneck(cin, planes, stride, need_ds)
    for m in rb.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(m.weight, 0.5, 1.5)
            torch.nn.init.normal_(m.bias, 0, 0.3)
    rb.train()
    mb.load_state_dict(rb.state_dict())
    h = _harness(mb).cuda().train()
    h.ensure_arena()
    x = torch.relu(torch.randn(2, cin, 12, 12)).requires_grad_(True)
    out_ref = rb(x)
    d_out = torch.randn(out_ref.shape)
    out_ref.backward(d_out)

    P = Plan(h, True, True)
    xd = _nhwc(x.detach())
    out, rec = mb.fwd(P, xd)
    check(_nchw(out), out_ref, "block out", 1e-4)
    P.begin_backward()
    G = GradSlots()
    G.put(out, _nhwc(d_out))
    prior = torch.randn(xd.shape, device="cuda")          # an earlier consumer already wrote into dx
    G.put(xd, prior.clone())
    mb.bwd(P, G, rec, out)
    P.join_side_stream()                                    # weight gradients are produced on a side stream
    check(_nchw(G.get(xd) - prior), x.grad, "block dx", 1e-3)
    gr = dict(rb.named_parameters())
    for (k, p), g in zip(h.named_parameters(), h.grad_views(P.garena)):
        check(g, gr[k[len("block."):]].grad, f"block grad {k}", 1e-3)


@pytest.mark.parametrize("cin,cskip,cout", [(512, 256, 256), (64, 64, 32), (32, 0, 16)])
def test_decoder_block_plan(pkg, cin, cskip, cout):
    from oracle import unet_ref as R
    from uda_aerial_semantic_segmentation_research_amd import unet as U
    from uda_aerial_semantic_segmentation_research_amd.engine import GradSlots, Plan
    torch.manual_seed(9)
    rb = R.DecoderBlockRef(cin, cskip, cout).train()
    mb = U.DecoderBlock(cin, cskip, cout)
    mb.load_state_dict(rb.state_dict())
    h = _harness(mb).cuda().train()
    h.ensure_arena()
    x = torch.randn(2, cin, 6, 5).requires_grad_(True)
    skip = torch.randn(2, cskip, 12, 10).requires_grad_(True) if cskip else None
    out_ref = rb(x, skip)
    d_out = torch.randn(out_ref.shape)
    out_ref.backward(d_out)
    P = Plan(h, True, True)
    xd, sd = _nhwc(x.detach()), (_nhwc(skip.detach()) if cskip else None)
    out, rec = mb.fwd(P, xd, sd)
    check(_nchw(out), out_ref, "decoder block out", 1e-4)
    P.begin_backward()
    G = GradSlots()
    G.put(out, _nhwc(d_out))
    mb.bwd(P, G, rec, out)
    P.join_side_stream()
    check(_nchw(G.get(xd)), x.grad, "decoder block dx", 1e-3)
    if cskip:
        check(_nchw(G.get(sd)), skip.grad, "decoder block dskip", 1e-3)
    gr = dict(rb.named_parameters())
    for (k, p), g in zip(h.named_parameters(), h.grad_views(P.garena)):
        check(g, gr[k[len("block."):]].grad, f"decoder grad {k}", 1e-3)


@pytest.mark.parametrize("name,dtype,phase", [("resnet18", torch.float32, False), ("resnet50", torch.float32, False),
                                              ("resnet18", torch.bfloat16, False), ("resnet18", torch.float32, True),
                                              ("resnet50", torch.float32, True)])
def test_fused_decoder_input_equals_materialised(pkg, name, dtype, phase, monkeypatch):
    """The decoder's cat([nearest_x2(x), skip]) gathered inside conv1 against the same network with the concatenation written by
    the stand-alone kernel.  phase=False (the nine-tap gather over the virtual concatenation): logits bit for bit (train and eval
    mode), gradients to atomics noise.  phase=True (round 5, the default on fp32: the up-sampled half as four 2x2 phase convolutions
    with pre-summed weights, csrc/conv_up_f32x3.hip): bit equality is NOT expected -- the pre-sums round once more and the sums run
    in another order -- train-mode logits agree to 1e-5, the head's gradients to 1e-5, the whole gradient by direction (a ReLU
    pre-activation within rounding distance of zero may change sign between two fp32 evaluations); eval mode does not use the
    phase kernels and stays bit-equal."""
    from oracle.adversarial_ref import synthetic_batch
    from uda_aerial_semantic_segmentation_research_amd import engine as E, unet as U
    monkeypatch.setattr(E, "USE_UP_PHASE", phase)
    from uda_aerial_semantic_segmentation_research_amd.losses import CrossEntropyLoss
    _, net = _pair(name, compute_dtype=dtype)
    x, y, _ = synthetic_batch(2, 64, 64, seed=4)
    xd, yd = x.cuda(), y.cuda()
    out = {}
    assert U.FUSE_UPCAT
    try:
        for fused in (False, True):
            U.FUSE_UPCAT = fused
            net.zero_grad()
            net.train()
            state = {k: v.clone() for k, v in net.state_dict().items()}
            net.debug_keep_tape = True
            logits = net(xd)
            kinds = [type(rec[2]).__name__ for blk, rec, _ in net._last_tape[1] if isinstance(blk, U.DecoderBlock)]
            CrossEntropyLoss()(logits, yd).backward()
            g = net._grad_arena.clone()
            net.load_state_dict(state)
            net.eval()
            with torch.no_grad():
                ev = net(xd)
            out[fused] = (logits.detach().clone(), g, ev.clone(), kinds)
    finally:
        U.FUSE_UPCAT = True
    assert out[False][3] == ["Tensor"] * 5
    assert out[True][3].count("UpCat") >= (4 if dtype == torch.bfloat16 else 5), out[True][3]
    assert torch.equal(out[True][2], out[False][2]), "eval-mode logits differ"
    if not phase:
        assert torch.equal(out[True][0], out[False][0]), "train-mode logits differ"
        e = ((out[True][1] - out[False][1]).abs().max() / out[False][1].abs().max()).item()
        assert e <= 1e-5, f"gradient arenas differ by {e:.3e}"
        return
    assert getattr(net, "_up_off", None), "the phase packings were not built"
    el = ((out[True][0] - out[False][0]).abs().max() / out[False][0].abs().max()).item()
    ga, gb = out[True][1].double(), out[False][1].double()
    cos = (ga @ gb / (ga.norm() * gb.norm())).item()
    o, nel, _ = net.entry_index()[(id(net.segmentation_head[0]), "weight")]
    eh = ((ga[o:o + nel] - gb[o:o + nel]).norm() / gb[o:o + nel].norm()).item()
    print(f"phase form vs materialised: logits {el:.3e} head gradient {eh:.3e} 1 - cos(all gradients) {1 - cos:.3e}")
    assert el <= 1e-5 and eh <= 1e-5 and cos >= 0.9995


def test_unet_bilinear_decoder_vs_oracle(pkg):
    """Unet(..., upsample="bilinear") -- the up-sampling mode north_star names (the reference's traced model uses nearest,
    SURVEY F5) -- against the oracle's bilinear mode: logits, loss, every gradient, eval mode."""
    from oracle.adversarial_ref import synthetic_batch
    from uda_aerial_semantic_segmentation_research_amd.losses import CrossEntropyLoss
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    ref, net = _pair("resnet18", upsample="bilinear")
    assert net.upsample == "bilinear" and all(b.upsample == "bilinear" for b in net.decoder.blocks)
    assert Unet("resnet18", classes=23, decoder_interpolation="bilinear").upsample == "bilinear"     # newer smp's keyword
    with pytest.raises(ValueError):
        Unet("resnet18", classes=23, upsample="bicubic")
    x, y, _ = synthetic_batch(2, 64, 96, seed=2)
    net.debug_keep_tape = True
    logits = net(x.cuda())
    loss = CrossEntropyLoss()(logits, y.cuda())
    loss.backward()
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    logits_ref = ref(x)
    ref.load_state_dict(state)
    check(logits, logits_ref, "bilinear logits")
    assert abs(loss.item() - torch.nn.functional.cross_entropy(logits_ref, y).item()) <= 1e-5 * abs(loss.item())
    grads_vs_oracle(net, ref, x, lambda o: torch.nn.functional.cross_entropy(o, y), "bilinear decoder")
    # and it is not the nearest network in disguise
    _, nearest = _pair("resnet18")
    with torch.no_grad():
        assert rel(nearest(x.cuda()), logits_ref) > 1e-2
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})     # same BatchNorm running statistics
    net.eval(), ref.eval()
    with torch.no_grad():
        check(net(x.cuda()), ref(x), "bilinear eval logits")


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
    x, _, _ = synthetic_batch(2, 64, 64, seed=5)
    wmap = torch.randn(2, 23, 64, 64)
    net.debug_keep_tape = True
    (net(x.cuda()) * wmap.cuda()).sum().backward()
    grads_vs_oracle(net, ref, x, lambda o: (o * wmap).sum(), "foreign-grad")


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


def test_unet_edge_shapes_and_errors(pkg):
    """Batch 1, non-square inputs, the smallest legal extent (32), the reference's shape error, dtype / device errors."""
    from oracle.unet_ref import UnetRef
    from uda_aerial_semantic_segmentation_research_amd.losses import CrossEntropyLoss
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    torch.manual_seed(3)
    ref = UnetRef("resnet18", classes=23).eval()
    net = Unet("resnet18", encoder_weights=None, in_channels=3, classes=23)
    net.load_state_dict(ref.state_dict())
    net = net.cuda().eval()
    for shape in ((1, 3, 32, 32), (1, 3, 64, 160), (3, 3, 96, 32)):
        x = torch.randn(*shape)
        with torch.no_grad():
            check(net(x.cuda()), ref(x), f"eval logits {shape}")
    # training on a non-square batch of one image: loss and the head's gradient
    net.train(), ref.train()
    x, y = torch.randn(1, 3, 64, 96), torch.randint(0, 23, (1, 64, 96))
    loss = CrossEntropyLoss()(net(x.cuda()), y.cuda())
    loss.backward()
    loss_ref = torch.nn.functional.cross_entropy(ref(x), y)
    loss_ref.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-4 * abs(loss_ref.item())
    check(net.segmentation_head[0].weight.grad, ref.segmentation_head[0].weight.grad, "head grad, 1 x 64 x 96")
    with pytest.raises(RuntimeError, match="divisible by 32"):
        net(torch.randn(1, 3, 48, 64, device="cuda"))
    with pytest.raises(ValueError):
        net(torch.randn(1, 4, 64, 64, device="cuda"))
    with pytest.raises(RuntimeError, match="GPU"):
        net(torch.randn(1, 3, 64, 64))
    # half-precision / double inputs are accepted like any float tensor (converted once)
    with torch.no_grad():
        net.eval()
        a = net(x.cuda())
        assert torch.equal(net(x.double().cuda()), a)


def test_bn_backward_reductions_in_the_dgrad_epilogue(pkg):
    """The BatchNorm-backward sums of single-consumer activations come out of the consumer's data-gradient epilogue
    (udaseg_conv2d_dgrad_bnreduce) instead of a separate pass: same gradients as with the stand-alone reduce kernel, and the
    fused form is really taken (r18 at 8 x 256 x 256: the conv2 data gradients of layer1 and of two decoder blocks; smaller
    layers run K-sliced launches whose atomic epilogue cannot carry the sums)."""
    from oracle.adversarial_ref import synthetic_batch
    from uda_aerial_semantic_segmentation_research_amd import engine as E
    from uda_aerial_semantic_segmentation_research_amd import kernels as K
    from uda_aerial_semantic_segmentation_research_amd.losses import CrossEntropyLoss
    _, net = _pair("resnet18")
    x, y, _ = synthetic_batch(8, 256, 256, seed=6)
    xd, yd = x.cuda(), y.cuda()
    calls = {"fused": 0, "reduce": 0}
    orig_f, orig_r, orig_h, orig_u, orig_n = (K.conv2d_dgrad_bnreduce, K.bn_bwd_reduce, K.conv2d_dgrad_frag, K.conv2d_dgrad_up,
                                               K.conv2d_dgrad_n16)

    def count_f(*a, **k):
        calls["fused"] += 1
        return orig_f(*a, **k)

    def count_h(*a, **k):          # the halo-resident data gradient (fp32: three-term split) with the sums in its epilogue
        calls["fused"] += int(k.get("bn") is not None)
        return orig_h(*a, **k)

    def count_u(*a, **k):          # round 5: the phase-form data gradient of a decoder conv1 (the previous block's output layer)
        calls["fused"] += int(k.get("bn") is not None)
        calls["up"] = calls.get("up", 0) + int(k.get("bn") is not None)
        return orig_u(*a, **k)

    def count_n(*a, **k):          # round 5: the sixteen-wide tile (decoder block 4 conv2's data gradient)
        calls["fused"] += int(k.get("bn") is not None)
        return orig_n(*a, **k)

    def count_r(*a, **k):
        calls["reduce"] += 1
        return orig_r(*a, **k)
    K.conv2d_dgrad_bnreduce, K.bn_bwd_reduce, K.conv2d_dgrad_frag, K.conv2d_dgrad_up, K.conv2d_dgrad_n16 = (count_f, count_r, count_h,
                                                                                                            count_u, count_n)
    out = {}
    try:
        for fused in (False, True):
            E.FUSE_BN_REDUCE = fused
            calls.update(fused=0, reduce=0, up=0)
            net.zero_grad()
            CrossEntropyLoss()(net(xd), yd).backward()
            out[fused] = (net._grad_arena.clone(), dict(calls))
    finally:
        E.FUSE_BN_REDUCE = True
        K.conv2d_dgrad_bnreduce, K.bn_bwd_reduce, K.conv2d_dgrad_frag, K.conv2d_dgrad_up, K.conv2d_dgrad_n16 = (orig_f, orig_r, orig_h,
                                                                                                                orig_u, orig_n)
    assert out[True][1]["up"] == 4, out[True][1]           # the outputs of decoder blocks 0..3: their only consumer is the next conv1
    assert out[False][1]["fused"] == 0 and out[False][1]["reduce"] == 30
    assert out[True][1]["fused"] >= 4 and out[True][1]["fused"] + out[True][1]["reduce"] == 30, out[True][1]
    e = ((out[True][0] - out[False][0]).abs().max() / out[False][0].abs().max()).item()
    print(f"BN-backward reductions fused into {out[True][1]['fused']} of 30 layers; gradient arenas differ by {e:.2e}")
    assert e <= 2e-5, e
