"""GPU parity, kernel level: every C-ABI entry point against a plain PyTorch fp32 CPU reference of the same op.

Tolerance: north_star's 1e-3 relative (fp32), applied norm-wise per tensor (max |diff| / max |ref|); most ops
land around 1e-6.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 1e-3


@pytest.fixture(scope="module")
def K():
    from uda_aerial_semantic_segmentation_research_amd import _lib, kernels
    _lib.require_gpu()
    kernels.ensure_workspace(torch.device("cuda", 0))
    return kernels


def dev(t):
    return t.to("cuda")


def nhwc(t):  # NCHW cpu -> NHWC cuda contiguous
    return t.permute(0, 2, 3, 1).contiguous().to("cuda")


def nchw(t):  # NHWC cuda -> NCHW cpu
    return t.detach().cpu().permute(0, 3, 1, 2).contiguous()


def relerr(got, ref):
    got, ref = got.double(), ref.double()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def assert_close(got, ref, what, rtol=RTOL):
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    e = relerr(got, ref)
    if e > rtol:
        d = (got.double() - ref.double()).abs()
        idx = torch.nonzero(d == d.max())[0].tolist()
        frac = (d > rtol * ref.abs().max()).double().mean().item()
        raise AssertionError(f"{what}: rel err {e:.3e} > {rtol:g}; worst at {idx} got {got[tuple(idx)]:.6g} "
                             f"ref {ref[tuple(idx)]:.6g}; {100 * frac:.2f}% elements off; shape {tuple(ref.shape)}")


def w_ohwi(w):  # [co,ci,kh,kw] cpu -> [co,kh,kw,ci] cuda
    return w.permute(0, 2, 3, 1).contiguous().to("cuda")


CONV_CASES = [
    # n, h, w, ci, co, k, stride, pad                         what it stands for
    (2, 16, 16, 64, 64, 3, 1, 1),     # layer1 3x3
    (2, 16, 16, 64, 128, 3, 2, 1),    # layerN.0.conv1 stride 2 (dgrad parity classes 1/2/2/4 taps)
    (2, 16, 16, 64, 128, 1, 2, 0),    # downsample 1x1/2 (three empty dgrad classes)
    (1, 8, 8, 256, 256, 3, 1, 1),     # deep layer -> 64x64 tile path
    (2, 32, 32, 4, 64, 7, 2, 3),      # stem 7x7/2 on the 3->4 padded image (Ci < 32: several taps per K-tile)
    (2, 32, 32, 4, 64, 4, 2, 1),      # discriminator conv0 4x4/2
    (1, 16, 16, 64, 128, 4, 2, 1),    # discriminator conv1
    (1, 24, 24, 32, 16, 3, 1, 1),     # decoder tail Co = 16 (N tile padded to 32)
    (1, 24, 24, 16, 24, 3, 1, 1),     # head 16 -> 23(+1)
    (1, 12, 20, 192, 64, 3, 1, 1),    # decoder conv1 after concat, non-square, Ci not a power of two
    (3, 9, 7, 8, 36, 3, 1, 1),        # ragged everything: M, N, K tails
    (1, 5, 5, 8, 8, 3, 2, 1),         # odd extent with stride 2
    (2, 16, 16, 64, 256, 1, 1, 0),    # bottleneck 1x1
    (2, 24, 24, 128, 64, 1, 1, 0), (1, 7, 9, 64, 128, 1, 1, 0),   # 1x1 at widths that are not multiples of the load pass / ragged M
    (1, 64, 64, 64, 64, 3, 1, 1),     # 128x64 tile path? (M = 4096 -> small) keep for coverage
    (2, 40, 20, 16, 16, 3, 1, 1),     # small-channel direct kernel, ragged 16x16 tiles
    (1, 33, 47, 32, 16, 3, 1, 1),     # direct kernel, two K groups, odd extents
    (2, 16, 16, 8, 32, 3, 1, 1),      # direct kernel, partial K group (ci = 8), two output tiles
    (1, 64, 64, 16, 24, 3, 1, 1),     # head shape through the direct kernel
    (4, 96, 96, 64, 64, 3, 1, 1),     # enough tiles for the unsliced launches: uniform-tap loop + buffer-store epilogues
    (2, 96, 96, 128, 32, 3, 1, 1),    # same through the 128x32 tile
    (2, 32, 32, 64, 128, 3, 1, 1),    # kernel-row wgrad tiles: two co tiles, image-border rows / columns in every K-step
    (3, 32, 64, 192, 64, 3, 1, 1),    # kernel-row wgrad tiles: three ci tiles, non-square, two K-steps per image row
    (5, 32, 32, 64, 64, 3, 1, 1),     # kernel-row wgrad tiles: pixel splits that end inside an image (5 images, 8-split lanes)
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[("n%d_%dx%d_ci%d_co%d_k%d_s%d_p%d" % c) for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(K, case):
    n, h, w, ci, co, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)
    bias = torch.randn(co, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, bias, stride=s, padding=p)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)

    d = K.conv_desc(n, h, w, ci, co, k, s, p)
    xd, wd, bd = nhwc(x), w_ohwi(wt), dev(bias)
    y = torch.full((n, d.ho, d.wo, co), float("nan"), device="cuda")
    K.conv2d_fwd(d, xd, wd, bd, y)
    assert_close(nchw(y), y_ref.detach(), "fwd")

    # fused bias + LeakyReLU epilogue and accumulate
    y2 = torch.ones((n, d.ho, d.wo, co), device="cuda")
    K.conv2d_fwd(d, xd, wd, bd, y2, act=1, slope=0.2, accumulate=True)
    assert_close(nchw(y2), F.leaky_relu(y_ref.detach(), 0.2) + 1.0, "fwd+leaky+acc")

    dyd = nhwc(dy)
    wtp = torch.empty((ci, k, k, co), device="cuda")
    K.pack_dgrad_weights(d, wd, wtp)
    assert torch.equal(wtp.cpu(), wt.permute(1, 2, 3, 0).contiguous()), "pack_dgrad_weights"
    dx = torch.full((n, h, w, ci), float("nan"), device="cuda")
    K.conv2d_dgrad(d, dyd, wtp, dx)
    assert_close(nchw(dx), xr.grad, "dgrad")
    dx2 = torch.full((n, h, w, ci), 2.0, device="cuda")
    K.conv2d_dgrad(d, dyd, wtp, dx2, accumulate=True)
    assert_close(nchw(dx2), xr.grad + 2.0, "dgrad+acc")

    dw = torch.full((co, k, k, ci), float("nan"), device="cuda")
    K.conv2d_wgrad(d, xd, dyd, dw)
    assert_close(dw.cpu().permute(0, 3, 1, 2), wr.grad, "wgrad")
    dw2 = torch.full((co, k, k, ci), 0.5, device="cuda")
    K.conv2d_wgrad(d, xd, dyd, dw2, accumulate=True)
    assert_close(dw2.cpu().permute(0, 3, 1, 2), wr.grad + 0.5, "wgrad+acc")


UPCAT_CASES = [
    # n, h, w (of the HALF-resolution a), ca, cb, co, dtype        stands for
    (2, 6, 10, 64, 32, 64, "fp32"),      # generic: 64 up-sampled + 32 skip channels, non-square, K-slices (few tiles)
    (2, 8, 8, 128, 64, 32, "fp32"),      # dec.3.conv1 shape class: Co = 32 -> 128x32 igemm tile, 32x128 wgrad tile
    (8, 24, 24, 64, 64, 64, "fp32"),     # enough tiles for unsliced launches and the buffer-store epilogues
    (1, 4, 4, 512, 256, 256, "fp32"),    # dec.0.conv1 channel counts at a tiny extent
    (2, 20, 12, 32, 0, 16, "fp32"),      # dec.4.conv1: no skip, small-channel direct kernels (halo read through the up-sampling)
    (1, 9, 7, 16, 0, 16, "fp32"),        # same, one K group, odd half-resolution extents (ragged 16x16 tiles)
    (2, 8, 8, 64, 0, 64, "fp32"),        # no skip through the implicit-GEMM kernels
    (2, 16, 16, 64, 64, 64, "fp32"),     # 32-pixel rows: the kernel-row weight-gradient tiles, one launch per source
    (1, 16, 32, 128, 64, 128, "fp32"),   # same, two co tiles, 64-pixel rows, up-sampled source with two ci tiles
    (2, 16, 16, 64, 0, 64, "fp32"),      # same, no skip
    (2, 8, 12, 64, 64, 64, "bf16"),      # bf16 storage
    (2, 8, 8, 128, 64, 64, "bf16"),
]


@pytest.mark.parametrize("case", UPCAT_CASES, ids=["n%d_%dx%d_ca%d_cb%d_co%d_%s" % c for c in UPCAT_CASES])
def test_conv_over_fused_upsample_concat(K, case):
    """conv(cat([nearest_x2(a), skip])) with the concatenation never written (udaseg_conv2d_fwd_upcat, _dgrad_split,
    _wgrad_part) against torch, and BIT FOR BIT (forward, data gradients) against the same library on the materialised
    concatenation: both walk the same K order."""
    n, h, w, ca, cb, co, dt = case
    bf = dt == "bf16"
    tol = 2 ** -7 if bf else RTOL
    g = torch.Generator().manual_seed(ca * 131 + cb * 7 + co)
    rnd = (lambda *s: torch.randn(*s, generator=g).bfloat16().float()) if bf else (lambda *s: torch.randn(*s, generator=g))
    a = rnd(n, ca, h, w).requires_grad_(True)
    skip = rnd(n, cb, 2 * h, 2 * w).requires_grad_(True) if cb else None
    ci = ca + cb
    wt = (rnd(co, ci, 3, 3) / math.sqrt(9 * ci)).requires_grad_(True)
    if bf:
        wt = (wt.detach().bfloat16().float()).requires_grad_(True)
    up = F.interpolate(a, scale_factor=2.0, mode="nearest")
    cat_ref = torch.cat([up, skip], dim=1) if cb else up
    y_ref = F.conv2d(cat_ref, wt, None, 1, 1)
    dy = rnd(*y_ref.shape)
    y_ref.backward(dy)

    cast = (lambda t: t.bfloat16()) if bf else (lambda t: t)
    ad, sd = cast(nhwc(a.detach())), (cast(nhwc(skip.detach())) if cb else None)
    wd, dyd = cast(w_ohwi(wt.detach())), cast(nhwc(dy))
    d = K.conv_desc(n, 2 * h, 2 * w, ci, co, 3, 1, 1)
    adt = torch.bfloat16 if bf else torch.float32
    assert K.upcat_fusable(ca, cb, co, adt)
    R = K.bn_replicas()
    # forward + BatchNorm statistics
    y = torch.full((n, 2 * h, 2 * w, co), float("nan"), device="cuda", dtype=adt)
    stats = torch.zeros(R * 2 * co, dtype=torch.float64, device="cuda")
    K.conv2d_fwd_upcat(d, ad, sd, wd, None, y, stats=stats)
    assert_close(nchw(y.float()), y_ref.detach(), "fused fwd", tol)
    cat = K.upsample2x_concat_fwd(ad, sd)
    y_mat = torch.empty_like(y)
    stats_mat = torch.zeros_like(stats)
    K.conv2d_fwd_bnstats(d, cat, wd, None, y_mat, stats_mat)
    assert torch.equal(y, y_mat), "fused forward differs from the forward on the materialised concatenation"
    s1, s2 = stats.view(R, 2, co).sum(0), stats_mat.view(R, 2, co).sum(0)
    assert torch.allclose(s1, s2, rtol=1e-9, atol=1e-9)
    # with bias + activation (the eval-mode folded-BatchNorm form)
    bias = torch.randn(co, generator=g)
    y2 = torch.empty_like(y)
    K.conv2d_fwd_upcat(d, ad, sd, wd, bias.cuda(), y2, act=1, slope=0.0)
    assert_close(nchw(y2.float()), torch.relu(y_ref.detach() + bias.view(1, -1, 1, 1)), "fused fwd + bias + relu", tol)
    # data gradient: two outputs
    wtp = torch.empty((ci, 3, 3, co), device="cuda", dtype=adt)
    if bf:
        wtp.copy_(wd.permute(3, 1, 2, 0))
    else:
        K.pack_dgrad_weights(d, wd, wtp)
    d_cat = torch.empty((n, 2 * h, 2 * w, ci), device="cuda", dtype=adt)
    K.conv2d_dgrad(d, dyd, wtp, d_cat)
    d_up = torch.full((n, 2 * h, 2 * w, ca), float("nan"), device="cuda", dtype=adt)
    if cb:
        d_skip = torch.full((n, 2 * h, 2 * w, cb), float("nan"), device="cuda", dtype=adt)
        K.conv2d_dgrad_split(d, dyd, wtp, d_up, d_skip)
        assert torch.equal(d_skip, d_cat[..., ca:]) and torch.equal(d_up, d_cat[..., :ca]), "split dgrad != slices of the plain one"
        assert_close(nchw(d_skip.float()), skip.grad, "d skip", 4 * tol if bf else tol)
    else:
        d_up = d_cat
    da = torch.empty((n, h, w, ca), device="cuda", dtype=adt)
    K.upsample2x_concat_bwd(d_up.contiguous(), da, None, ca, 0)
    assert_close(nchw(da.float()), a.grad, "d a (2x2 sum of the up-sampled gradient)", 4 * tol if bf else tol)
    # weight gradient: one launch per source, accumulated onto a zeroed gradient
    dw = torch.zeros((co, 3, 3, ci), device="cuda")
    K.conv2d_wgrad_part(d, ad, 0, True, dyd, dw, True)
    if cb:
        K.conv2d_wgrad_part(d, sd, ca, False, dyd, dw, True)
    assert_close(dw.cpu().permute(0, 3, 1, 2), wt.grad, "wgrad by source", 4 * tol if bf else tol)
    dw_mat = torch.zeros_like(dw)
    K.conv2d_wgrad(d, cat, dyd, dw_mat, accumulate=True)
    assert relerr(dw.cpu(), dw_mat.cpu()) < 1e-5, "wgrad by source vs wgrad on the materialised concatenation"


@pytest.mark.parametrize("n,h,w,ca,cb,dt", [(2, 5, 6, 32, 16, "fp32"), (1, 1, 1, 8, 0, "fp32"), (2, 1, 7, 4, 4, "fp32"),
                                            (1, 9, 2, 64, 64, "fp32"), (2, 6, 5, 32, 16, "bf16"), (1, 3, 3, 8, 0, "bf16")])
def test_bilinear_upsample_concat_vs_interpolate(K, n, h, w, ca, cb, dt):
    """cat(F.interpolate(a, scale_factor=2, mode='bilinear', align_corners=False), skip), forward and backward (SURVEY 8(c)
    item 5: nearest vs bilinear micro-vectors), including 1-pixel extents where both neighbours clamp onto the same row."""
    bf = dt == "bf16"
    g = torch.Generator().manual_seed(h * 100 + w)
    rnd = (lambda *s: torch.randn(*s, generator=g).bfloat16().float()) if bf else (lambda *s: torch.randn(*s, generator=g))
    a = rnd(n, ca, h, w).requires_grad_(True)
    skip = rnd(n, cb, 2 * h, 2 * w).requires_grad_(True) if cb else None
    up = F.interpolate(a, scale_factor=2.0, mode="bilinear", align_corners=False)
    ref = torch.cat([up, skip], 1) if cb else up
    dout = rnd(*ref.shape)
    ref.backward(dout)
    cast = (lambda t: t.bfloat16()) if bf else (lambda t: t)
    tol = 2 ** -7 if bf else 1e-6
    out = K.upsample2x_bilinear_concat_fwd(cast(nhwc(a.detach())), cast(nhwc(skip.detach())) if cb else None)
    assert_close(nchw(out.float()), ref.detach(), "bilinear fwd", tol)
    da = torch.full((n, h, w, ca), float("nan"), device="cuda", dtype=out.dtype)
    ds = torch.full((n, 2 * h, 2 * w, cb), float("nan"), device="cuda", dtype=out.dtype) if cb else None
    K.upsample2x_bilinear_concat_bwd(cast(nhwc(dout)), da, ds, ca, cb)
    assert_close(nchw(da.float()), a.grad, "bilinear bwd", 4 * tol)
    if cb:
        assert torch.equal(nchw(ds.float()), skip.grad)
    # accumulate forms
    da2 = torch.ones_like(da)
    K.upsample2x_bilinear_concat_bwd(cast(nhwc(dout)), da2, None, ca, cb, accumulate_da=True)
    assert_close(nchw(da2.float()), a.grad + 1.0, "bilinear bwd + acc", 8 * tol if bf else 4 * tol)


def test_fused_upcat_rejects_what_it_cannot_do(K):
    d = K.conv_desc(1, 8, 8, 48, 64, 3, 1, 1)
    a, skip = torch.zeros(1, 4, 4, 24, device="cuda"), torch.zeros(1, 8, 8, 24, device="cuda")
    w, y = torch.zeros(64, 3, 3, 48, device="cuda"), torch.zeros(1, 8, 8, 64, device="cuda")
    assert not K.upcat_fusable(24, 24, 64, torch.float32)
    with pytest.raises(RuntimeError, match="multiples of the K-tile"):
        K.conv2d_fwd_upcat(d, a, skip, w, None, y)
    with pytest.raises(RuntimeError, match="up_ca"):
        K.conv2d_fwd_upcat(d, torch.zeros(1, 4, 4, 48, device="cuda"), skip, w, None, y)      # ca == ci but a skip is given
    dw = torch.zeros(64, 3, 3, 48, device="cuda")
    with pytest.raises(RuntimeError, match="accumulated"):
        K.conv2d_wgrad_part(d, a, 0, True, y, dw, accumulate=False)


@pytest.mark.parametrize("case", [(2, 24, 24, 64, 64, 3, 1, 1), (2, 20, 36, 16, 16, 3, 1, 1), (4, 40, 40, 32, 16, 3, 1, 1),
                                  (1, 8, 8, 256, 256, 3, 1, 1), (2, 32, 32, 4, 64, 4, 2, 1), (8, 96, 96, 64, 128, 3, 1, 1)])
def test_conv_fwd_with_fused_bn_statistics(K, case):
    """conv epilogue statistics == what the separate bn_stats pass adds (igemm, small-channel and K-sliced paths)."""
    n, h, w, ci, co, k, s, p = case
    g = torch.Generator().manual_seed(ci * co)
    x = torch.randn(n, h, w, ci, generator=g).cuda()
    wt = (torch.randn(co, k, k, ci, generator=g) / math.sqrt(ci * k * k)).cuda()
    bias = torch.randn(co, generator=g).cuda()
    d = K.conv_desc(n, h, w, ci, co, k, s, p)
    R = K.bn_replicas()
    for b in (None, bias):
        y = torch.empty((n, d.ho, d.wo, co), device="cuda")
        st = torch.zeros(R * 2 * co, dtype=torch.float64, device="cuda")
        K.conv2d_fwd_bnstats(d, x, wt, b, y, st)
        y2 = torch.empty_like(y)
        K.conv2d_fwd(d, x, wt, b, y2)
        assert torch.equal(y, y2) or relerr(y.cpu(), y2.cpu()) < 1e-6
        tot = st.view(R, 2, co).sum(0).cpu()
        yd = y.double().reshape(-1, co).cpu()
        assert_close(tot[0], yd.sum(0), "fused sum", 1e-5)
        assert_close(tot[1], (yd * yd).sum(0), "fused sum of squares", 1e-5)


def test_conv_identity_weights_asymmetric(K):
    """A = I check with an asymmetric operand: catches a transposed C-write or a swapped fragment map."""
    n, h, w, c = 1, 8, 8, 64
    x = torch.arange(n * h * w * c, dtype=torch.float32).reshape(n, h, w, c) * 1e-3
    wt = torch.zeros(c, 1, 1, c)
    for o in range(c):
        wt[o, 0, 0, (o + 5) % c] = 1.0 + o          # permutation + per-output scale: asymmetric
    d = K.conv_desc(n, h, w, c, c, 1, 1, 0)
    y = torch.empty((n, h, w, c), device="cuda")
    K.conv2d_fwd(d, x.cuda(), wt.cuda(), None, y)
    ref = torch.stack([x[..., (o + 5) % c] * (1.0 + o) for o in range(c)], dim=-1)
    # since round 4 this launch runs on the bf16 matrix pipe (three-term split, conv_igemm_kernel X3): x * w arrives as three exact
    # partial products whose sum the MFMA adder rounds in its own order -- within one unit in the last place of the fp32 product,
    # which still separates every (pixel, channel) from its neighbours (the values are distinct multiples of 1e-3 x (1 + o))
    err = ((y.cpu() - ref).abs() / ref.abs().clamp_min(1e-30)).max().item()
    assert err <= 2.0 ** -22, err


def test_conv_large_tiles(K):
    """Shapes big enough to take the 128x128 / 128x64 tile paths and split-K wgrad; reference = fp64 on CPU subsample."""
    for (n, h, w, ci, co) in [(8, 96, 96, 64, 64), (8, 80, 80, 64, 128)]:
        g = torch.Generator().manual_seed(7)
        x = torch.randn(n, ci, h, w, generator=g)
        wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
        d = K.conv_desc(n, h, w, ci, co, 3, 1, 1)
        y = torch.empty((n, h, w, co), device="cuda")
        K.conv2d_fwd(d, nhwc(x), w_ohwi(wt), None, y)
        y_ref = F.conv2d(x, wt, None, padding=1)
        assert_close(nchw(y), y_ref, f"fwd big {ci}->{co}")
        dy = torch.randn(y_ref.shape, generator=g)
        dw = torch.empty((co, 3, 3, ci), device="cuda")
        K.conv2d_wgrad(d, nhwc(x), nhwc(dy), dw)
        dw_ref = torch.nn.grad.conv2d_weight(x, wt.shape, dy, padding=1)
        assert_close(dw.cpu().permute(0, 3, 1, 2), dw_ref, f"wgrad big {ci}->{co}")
        wtp = torch.empty((ci, 3, 3, co), device="cuda")
        K.pack_dgrad_weights(d, w_ohwi(wt), wtp)
        dx = torch.empty((n, h, w, ci), device="cuda")
        K.conv2d_dgrad(d, nhwc(dy), wtp, dx)
        dx_ref = torch.nn.grad.conv2d_input(x.shape, wt, dy, padding=1)
        assert_close(nchw(dx), dx_ref, f"dgrad big {ci}->{co}")


@pytest.mark.parametrize("c", [16, 64, 24, 512, 2048])
@pytest.mark.parametrize("act,slope,with_res", [(0, 0.0, False), (1, 0.0, True), (1, 0.2, False)])
def test_bn_train_fwd_bwd(K, c, act, slope, with_res):
    n, h, w = (2, 6, 10) if c > 64 else (4, 40, 52)       # the wide case spans many blocks -> all replicas in use
    g = torch.Generator().manual_seed(c + act)
    x = (torch.randn(n, c, h, w, generator=g) * 2 + 0.5).requires_grad_(True)
    res = torch.randn(n, c, h, w, generator=g).requires_grad_(True) if with_res else None
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g))
        bn.running_mean.copy_(torch.randn(c, generator=g))
        bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    bn.train()
    u = bn(x)
    if with_res:
        u = u + res
    z_ref = F.leaky_relu(u, slope) if act else u
    dz = torch.randn(z_ref.shape, generator=g)
    z_ref.backward(dz)

    yd = nhwc(x.detach())
    sums = torch.zeros(2 * c * K.bn_replicas(), dtype=torch.float64, device="cuda")
    K.bn_stats(yd, sums)
    z = torch.empty_like(yd)
    rm, rv = dev(rm0), dev(rv0)
    sm, sr = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    gam, bet = dev(bn.weight.detach()), dev(bn.bias.detach())
    resd = nhwc(res.detach()) if with_res else None
    K.bn_apply(yd, sums, gam, bet, resd, z, bn.eps, bn.momentum, rm, rv, sm, sr, act, slope)
    assert_close(nchw(z), z_ref.detach(), "bn fwd")
    assert_close(rm.cpu(), bn.running_mean, "running_mean", 1e-5)
    assert_close(rv.cpu(), bn.running_var, "running_var", 1e-5)

    bs = torch.zeros(2 * c * K.bn_replicas(), dtype=torch.float64, device="cuda")
    dzd = nhwc(dz)
    K.bn_bwd_reduce(dzd, z, yd, sm, sr, bs, act, slope)
    dy = torch.empty_like(yd)
    dres = torch.empty_like(yd) if with_res else None
    dg, db = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    K.bn_bwd_apply(dzd, z, yd, sm, sr, gam, bs, dy, dres, dg, db, act, slope)
    assert_close(nchw(dy), x.grad, "bn dx")
    assert_close(dg.cpu(), bn.weight.grad, "bn dgamma")
    assert_close(db.cpu(), bn.bias.grad, "bn dbeta")
    if with_res:
        assert_close(nchw(dres), res.grad, "bn dres")
    elif act:
        # the z-less backward (no residual): the activation's argument is re-evaluated from y, gamma, beta with the same
        # fused multiply-add as the forward, so the mask is the same; the f64 sums differ only by atomic ordering
        bs2 = torch.zeros_like(bs)
        K.bn_bwd_reduce(dzd, None, yd, sm, sr, bs2, act, slope, gamma=gam, beta=bet)
        fold = lambda t: t.view(K.bn_replicas(), -1).sum(0)
        assert torch.allclose(fold(bs2), fold(bs), rtol=1e-12, atol=1e-12)
        dy2, dg2, db2 = torch.empty_like(yd), torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
        K.bn_bwd_apply(dzd, None, yd, sm, sr, gam, bs, dy2, None, dg2, db2, act, slope, beta=bet)
        assert torch.equal(dy2, dy) and torch.equal(dg2, dg) and torch.equal(db2, db)      # same sums in -> same bits out
    # eval mode
    bn.eval()
    with torch.no_grad():
        ze_ref = bn(x.detach())
        if with_res:
            ze_ref = ze_ref + res.detach()
        ze_ref = F.leaky_relu(ze_ref, slope) if act else ze_ref
    ze = torch.empty_like(yd)
    K.bn_apply_eval(yd, gam, bet, rm, rv, resd, ze, bn.eps, act, slope)
    assert_close(nchw(ze), ze_ref, "bn eval")


def test_act_bwd_and_channel_sum(K):
    g = torch.Generator().manual_seed(3)
    z = torch.randn(2, 5, 7, 24, generator=g)
    dz = torch.randn(2, 5, 7, 24, generator=g)
    dy = torch.empty(2, 5, 7, 24, device="cuda")
    K.act_bwd(dz.cuda(), z.cuda(), dy, 1, 0.2)
    assert_close(dy.cpu(), dz * torch.where(z > 0, 1.0, 0.2), "act_bwd")
    out = torch.full((24,), 3.0, device="cuda")
    K.channel_sum(dz.cuda(), out, accumulate=False)
    assert_close(out.cpu(), dz.sum(dim=(0, 1, 2)), "channel_sum", 1e-5)
    K.channel_sum(dz.cuda(), out, accumulate=True)
    assert_close(out.cpu(), 2 * dz.sum(dim=(0, 1, 2)), "channel_sum acc", 1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pack_dgrad_batched_tiles(K, dtype):
    """One launch repacks every convolution of a network for the data gradient: w[co][t][ci] -> wt[ci][t][co] as LDS tile
    transposes.  Ragged tiles (24, 36, 4 channels), a 7x7 and a 1x1 kernel, a full 512 x 512 x 9 layer; bit-exact."""
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 49, 4), (24, 9, 16), (16, 9, 36), (128, 1, 64), (512, 9, 512), (32, 16, 8), (64, 9, 64)]
    ws = [torch.randn(co, t, ci, generator=g) for co, t, ci in shapes]
    arena = torch.cat([w.reshape(-1) for w in ws]).cuda()
    rows, off = [], 0
    for (co, t, ci) in shapes:
        rows.append([off, off, co, t, ci])
        off += co * t * ci
    table = torch.tensor(rows, dtype=torch.int32, device="cuda")
    packed = torch.full((off,), float("nan"), device="cuda").to(dtype)
    K.pack_dgrad_batched(arena, packed, table)
    torch.cuda.synchronize()
    for (o, _, co, t, ci), w in zip(rows, ws):
        want = w.permute(2, 1, 0).contiguous().to(dtype)
        got = packed[o:o + co * t * ci].view(ci, t, co).cpu()
        assert torch.equal(got, want), (co, t, ci)


@pytest.mark.parametrize("shape,dtype", [((8, 256, 256, 64), torch.bfloat16), ((8, 64, 64, 256), torch.bfloat16),
                                         ((4, 128, 128, 128), torch.float32), ((2, 96, 96, 24), torch.float32),
                                         ((2, 32, 32, 1024), torch.float32)])
def test_channel_sum_replica_path(K, shape, dtype):
    """Bias gradients of the discriminator's strided convs (adversarial_trainer.py:85-114 backward): tensors large enough for
    the replica + fold path (csrc/common.h CHSUM_*), both storage types, overwrite and accumulate, non-zero mean so a lost
    replica would show."""
    g = torch.Generator().manual_seed(11)
    dz = (torch.randn(*shape, generator=g) + 0.25).to(dtype).cuda()
    ref = dz.double().sum(dim=(0, 1, 2)).float().cpu()
    out = torch.full((shape[-1],), 7.0, device="cuda")
    K.channel_sum(dz, out, accumulate=False)
    assert_close(out.cpu(), ref, "channel_sum", 2e-5)
    K.channel_sum(dz, out, accumulate=True)
    assert_close(out.cpu(), 2 * ref, "channel_sum acc", 2e-5)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        K.channel_sum(dz, out, accumulate=False, st=side.cuda_stream)
    side.synchronize()
    assert_close(out.cpu(), ref, "channel_sum on a side stream", 2e-5)


@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (1, 8, 7, 9), (2, 16, 2, 2)])
def test_maxpool(K, shape):
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn(*shape, generator=g)).requires_grad_(True)   # many exact ties at 0, like post-ReLU
    y_ref = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy)
    y, idx = K.maxpool_fwd(nhwc(x.detach()))
    assert torch.equal(nchw(y), y_ref.detach())
    dx = torch.full((shape[0], shape[2], shape[3], shape[1]), float("nan"), device="cuda")
    K.maxpool_bwd(nhwc(dy), idx, dx)
    assert_close(nchw(dx), x.grad, "maxpool bwd (tie rule)", 1e-6)


@pytest.mark.parametrize("ca,cb", [(32, 0), (64, 64), (512, 256)])
def test_upsample_concat(K, ca, cb):
    g = torch.Generator().manual_seed(2)
    a = torch.randn(2, ca, 5, 6, generator=g).requires_grad_(True)
    s = torch.randn(2, cb, 10, 12, generator=g).requires_grad_(True) if cb else None
    up = F.interpolate(a, scale_factor=2.0, mode="nearest")
    ref = torch.cat([up, s], dim=1) if cb else up
    dout = torch.randn(ref.shape, generator=g)
    ref.backward(dout)
    out = K.upsample2x_concat_fwd(nhwc(a.detach()), nhwc(s.detach()) if cb else None)
    assert torch.equal(nchw(out), ref.detach())
    da = torch.empty((2, 5, 6, ca), device="cuda")
    ds = torch.empty((2, 10, 12, cb), device="cuda") if cb else None
    K.upsample2x_concat_bwd(nhwc(dout), da, ds, ca, cb)
    assert_close(nchw(da), a.grad, "upcat da", 1e-6)
    if cb:
        assert torch.equal(nchw(ds), s.grad)


def test_nchw_to_nhwc(K):
    x = torch.randn(3, 3, 10, 14)
    y = K.nchw_to_nhwc(x.cuda(), 4)
    assert y.shape == (3, 10, 14, 4)
    assert torch.equal(y[..., :3].cpu(), x.permute(0, 2, 3, 1))
    assert float(y[..., 3].abs().max()) == 0.0


@pytest.mark.parametrize("classes,ldc", [(23, 24), (4, 4), (17, 20), (40, 40)])
def test_cross_entropy(K, classes, ldc):
    from uda_aerial_semantic_segmentation_research_amd import _lib
    g = torch.Generator().manual_seed(5)
    n, h, w = 2, 9, 11
    logits = (torch.randn(n, classes, h, w, generator=g) * 3).requires_grad_(True)
    tgt = torch.randint(0, classes, (n, h, w), generator=g)
    loss_ref = F.cross_entropy(logits, tgt)
    (loss_ref * 0.7).backward()
    buf = torch.full((n, h, w, ldc), 7.0, device="cuda")       # pad channels hold junk on purpose
    buf[..., :classes] = logits.detach().permute(0, 2, 3, 1).cuda()
    pixels = n * h * w
    lse = torch.empty(pixels, device="cuda")
    partials = torch.empty(_lib.load().udaseg_ce_partials(), dtype=torch.float64, device="cuda")
    loss = torch.empty((), device="cuda")
    K.ce_fwd(buf, tgt.cuda(), pixels, classes, ldc, lse, partials, loss)
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    gout = torch.tensor(0.7, device="cuda")
    dl = torch.full((n, h, w, ldc), float("nan"), device="cuda")
    if ldc <= 32:
        parts = torch.empty(_lib.load().udaseg_ce_partials() * ldc, device="cuda")
        colsum = torch.full((ldc,), float("nan"), device="cuda")
        K.ce_bwd(buf, tgt.cuda(), lse, gout, pixels, classes, ldc, dl, parts, colsum)
        assert_close(colsum[:classes].cpu(), logits.grad.sum(dim=(0, 2, 3)), "ce bwd column sums (head bias grad)", 1e-4)
        dl2 = torch.full((n, h, w, ldc), float("nan"), device="cuda")
        K.ce_bwd(buf, tgt.cuda(), lse, gout, pixels, classes, ldc, dl2)      # without the fused column sums
        assert torch.equal(dl2, dl)
    else:
        K.ce_bwd(buf, tgt.cuda(), lse, gout, pixels, classes, ldc, dl)
    assert_close(dl[..., :classes].cpu().permute(0, 3, 1, 2), logits.grad, "ce bwd", 1e-5)
    if ldc > classes:
        assert float(dl[..., classes:].abs().max()) == 0.0


@pytest.mark.parametrize("classes,ldc,shape", [(23, 24, (2, 9, 11)), (4, 4, (1, 5, 7)), (17, 20, (3, 40, 33)), (23, 24, (8, 128, 128)), (32, 32, (2, 16, 16))])
def test_cross_entropy_forward_and_backward_in_one_pass(K, classes, ldc, shape):
    """udaseg_ce_fwd_bwd against the two passes it replaces: loss, gradient (upstream gradient 1) and the head's bias gradient bit for
    bit; udaseg_scale_unless_one leaves them untouched for an upstream gradient of exactly 1 and scales them otherwise; through the
    autograd function the fused and the two-pass route give the same bits for loss.backward()."""
    from uda_aerial_semantic_segmentation_research_amd import _lib, losses
    g = torch.Generator().manual_seed(sum(shape) + classes)
    n, h, w = shape
    logits = torch.randn(n, classes, h, w, generator=g) * 3
    tgt = torch.randint(0, classes, (n, h, w), generator=g).cuda()
    buf = torch.full((n, h, w, ldc), 7.0, device="cuda")
    buf[..., :classes] = logits.permute(0, 2, 3, 1).cuda()
    pixels = n * h * w
    P = _lib.load().udaseg_ce_partials()
    lse, partials, loss = torch.empty(pixels, device="cuda"), torch.empty(P, dtype=torch.float64, device="cuda"), torch.empty((), device="cuda")
    K.ce_fwd(buf, tgt, pixels, classes, ldc, lse, partials, loss)
    one = torch.ones((), device="cuda")
    dl, parts, colsum = torch.full((n, h, w, ldc), float("nan"), device="cuda"), torch.empty(P * ldc, device="cuda"), torch.empty(ldc, device="cuda")
    K.ce_bwd(buf, tgt, lse, one, pixels, classes, ldc, dl, parts, colsum)
    loss2, dl2, colsum2 = torch.empty((), device="cuda"), torch.full((n, h, w, ldc), float("nan"), device="cuda"), torch.empty(ldc, device="cuda")
    K.ce_fwd_bwd(buf, tgt, pixels, classes, ldc, torch.empty(P, dtype=torch.float64, device="cuda"), loss2, dl2, torch.empty(P * ldc, device="cuda"), colsum2)
    assert torch.equal(loss2, loss) and torch.equal(dl2, dl) and torch.equal(colsum2, colsum)
    K.scale_unless_one(dl2, one, colsum2)
    assert torch.equal(dl2, dl) and torch.equal(colsum2, colsum)
    gout = torch.tensor(0.37, device="cuda")
    K.scale_unless_one(dl2, gout, colsum2)
    dl3, colsum3 = torch.empty_like(dl), torch.empty(ldc, device="cuda")
    K.ce_bwd(buf, tgt, lse, gout, pixels, classes, ldc, dl3, parts, colsum3)
    assert torch.allclose(dl2, dl3, rtol=2e-7, atol=0) and torch.allclose(colsum2, colsum3, rtol=1e-5, atol=1e-12)
    # the module, both routes
    crit = losses.CrossEntropyLoss()
    res = {}
    for fused in (True, False):
        losses.FUSE_CE_BACKWARD = fused
        try:
            x = logits.cuda().requires_grad_(True)
            l = crit(x, tgt)
            l.backward()
            res[fused] = (l.detach().clone(), x.grad.clone())
        finally:
            losses.FUSE_CE_BACKWARD = True
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    # a scaled loss and a second backward through a retained graph
    x = logits.cuda().requires_grad_(True)
    l = crit(x, tgt)
    (l * 0.5).backward(retain_graph=True)
    g1 = x.grad.clone()
    x.grad = None
    (l * 0.5).backward()
    assert torch.allclose(x.grad, g1, rtol=1e-6, atol=0) and torch.allclose(g1, res[False][1] * 0.5, rtol=1e-6, atol=0)
    with torch.no_grad():
        assert torch.equal(crit(logits.cuda(), tgt), res[False][0])


def test_cross_entropy_extreme_logits(K):
    """Large-magnitude logits: the max-shifted log-sum-exp must not overflow (torch's own behaviour)."""
    from uda_aerial_semantic_segmentation_research_amd import _lib
    logits = torch.tensor([[[[80.0]], [[-90.0]], [[0.0]], [[79.0]]]])      # [1,4,1,1]
    tgt = torch.tensor([[[1]]])
    ref = F.cross_entropy(logits, tgt)
    buf = logits.permute(0, 2, 3, 1).contiguous().cuda()
    lse = torch.empty(1, device="cuda")
    partials = torch.empty(_lib.load().udaseg_ce_partials(), dtype=torch.float64, device="cuda")
    loss = torch.empty((), device="cuda")
    K.ce_fwd(buf, tgt.cuda(), 1, 4, 4, lse, partials, loss)
    assert math.isfinite(loss.item()) and abs(loss.item() - ref.item()) < 1e-4 * ref.item()


def test_gap_linear_sigmoid(K):
    g = torch.Generator().manual_seed(9)
    n, c, h, w = 3, 512, 4, 6
    z = torch.randn(n, c, h, w, generator=g).requires_grad_(True)
    lin = torch.nn.Linear(c, 1)
    p_ref = torch.sigmoid(lin(F.adaptive_avg_pool2d(z, 1).flatten(1)))
    dp = torch.randn(n, 1, generator=g)
    p_ref.backward(dp)
    wd, bd = lin.weight.detach().reshape(-1).cuda(), lin.bias.detach().cuda()
    p, pooled = K.gap_linear_sigmoid_fwd(nhwc(z.detach()), wd, bd)
    assert_close(p.cpu(), p_ref.detach(), "gap/linear/sigmoid fwd", 1e-5)
    dz = torch.empty((n, h, w, c), device="cuda")
    dw, db = torch.empty(c, device="cuda"), torch.empty(1, device="cuda")
    K.gap_linear_sigmoid_bwd(dp.cuda(), p, pooled, wd, dz, dw, db)
    assert_close(nchw(dz), z.grad, "gap bwd dz", 1e-5)
    assert_close(dw.cpu(), lin.weight.grad.reshape(-1), "gap bwd dw", 1e-5)
    assert_close(db.cpu(), lin.bias.grad, "gap bwd db", 1e-5)


def test_bce_with_logits_known_answers(K, golden_dir):
    """The reference's AdversarialLoss on its fixed [4,1] inputs (vectors made by importing src.models.losses)."""
    import os
    import numpy as np
    gold = np.load(os.path.join(golden_dir, "adversarial_ref.npz"))
    p_s, p_t = torch.from_numpy(gold["ka/p_s"]).cuda(), torch.from_numpy(gold["ka/p_t"]).cuda()
    loss = torch.empty((), device="cuda")
    K.bce_logits_fwd(p_s, 1.0, 0.5, loss, accumulate=False)
    K.bce_logits_fwd(p_t, 0.0, 0.5, loss, accumulate=True)
    assert abs(loss.item() - float(gold["ka/d_loss"])) < 1e-6
    K.bce_logits_fwd(p_t, 1.0, 0.001, loss, accumulate=False)
    assert abs(loss.item() - float(gold["ka/g_loss"])) < 1e-9
    x = p_t.cpu().clone().requires_grad_(True)
    (0.001 * F.binary_cross_entropy_with_logits(x, torch.ones_like(x)) * 2.0).backward()
    dx = torch.empty(4, 1, device="cuda")
    K.bce_logits_bwd(p_t, 1.0, 0.001, torch.tensor(2.0, device="cuda"), dx)
    assert_close(dx.cpu(), x.grad, "bce bwd", 1e-5)


@pytest.mark.parametrize("count", [4096, 1027])
def test_adam_matches_torch(K, count):
    g = torch.Generator().manual_seed(11)
    p0 = torch.randn(count, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-3)
    pd = p0.clone().cuda()
    m, v = torch.zeros(count, device="cuda"), torch.zeros(count, device="cuda")
    for t in range(1, 4):
        grad = torch.randn(count, generator=g) * (10.0 ** (t - 2))
        p_ref.grad = grad.clone()
        opt.step()
        K.adam_flat(pd, grad.cuda(), m, v, count, 1e-3, 0.9, 0.999, 1e-8, 1 - 0.9 ** t, 1 - 0.999 ** t)
    assert_close(pd.cpu(), p_ref.detach(), "adam params", 1e-6)


def test_bad_arguments_raise(K):
    d = K.conv_desc(1, 8, 8, 6, 8, 3, 1, 1)    # ci not a multiple of 4
    t = torch.empty(1024, device="cuda")
    with pytest.raises(RuntimeError, match="multiples of 4"):
        K.conv2d_fwd(d, t, t, None, t)


def test_argmax_confusion_matrix(K):
    g = torch.Generator().manual_seed(21)
    n, c, h, w, ldc = 3, 23, 17, 19, 24
    logits = torch.randn(n, c, h, w, generator=g)
    logits[0, 5, :, :4] = logits[0, 2, :, :4] = 9.0            # exact ties: first maximal index wins (torch.argmax)
    tgt = torch.randint(0, c, (n, h, w), generator=g)
    buf = torch.full((n, h, w, ldc), 50.0, device="cuda")       # pad channel holds a LARGER value: must be ignored
    buf[..., :c] = logits.permute(0, 2, 3, 1).cuda()
    cm = torch.zeros(c * c, dtype=torch.int64, device="cuda")
    pred = torch.empty(n * h * w, dtype=torch.int64, device="cuda")
    K.argmax_confusion(buf, tgt.cuda().reshape(-1), n * h * w, c, ldc, cm, pred)
    pr = logits.argmax(dim=1)
    assert torch.equal(pred.cpu().view(n, h, w), pr)
    ref = torch.bincount((tgt * c + pr).reshape(-1), minlength=c * c)     # src/analysis/metrics.py:17-29
    assert torch.equal(cm.cpu(), ref)


def test_bn_fold_and_fused_inference_conv(K):
    g = torch.Generator().manual_seed(22)
    n, h, w, ci, co = 2, 12, 14, 64, 128
    x = torch.randn(n, ci, h, w, generator=g)
    res = torch.randn(n, co, h, w, generator=g)
    conv = torch.nn.Conv2d(ci, co, 3, padding=1, bias=True)
    bn = torch.nn.BatchNorm2d(co).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(co, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(co, generator=g))
        bn.running_mean.copy_(torch.randn(co, generator=g))
        bn.running_var.copy_(torch.rand(co, generator=g) + 0.5)
        ref = F.relu(bn(conv(x)) + res)
    wd = w_ohwi(conv.weight.detach())
    wf, bf = torch.empty_like(wd), torch.empty(co, device="cuda")
    K.bn_fold(wd, dev(conv.bias.detach()), dev(bn.weight.detach()), dev(bn.bias.detach()), dev(bn.running_mean),
              dev(bn.running_var), bn.eps, wf, bf)
    d = K.conv_desc(n, h, w, ci, co, 3, 1, 1)
    y = torch.empty((n, h, w, co), device="cuda")
    K.conv2d_fwd_fused(d, nhwc(x), wf, bf, nhwc(res), y, act=1, slope=0.0)
    assert_close(nchw(y), ref, "folded conv+bn+add+relu", 1e-4)
    # small-channel direct kernel path with residual
    conv2 = torch.nn.Conv2d(16, 16, 3, padding=1, bias=False)
    x2, r2 = torch.randn(1, 16, 20, 20, generator=g), torch.randn(1, 16, 20, 20, generator=g)
    d2 = K.conv_desc(1, 20, 20, 16, 16, 3, 1, 1)
    y2 = torch.empty((1, 20, 20, 16), device="cuda")
    K.conv2d_fwd_fused(d2, nhwc(x2), w_ohwi(conv2.weight.detach()), None, nhwc(r2), y2, act=1, slope=0.1)
    with torch.no_grad():
        assert_close(nchw(y2), F.leaky_relu(conv2(x2) + r2, 0.1), "small conv + residual + leaky", 1e-4)


@pytest.mark.parametrize("case", [(8, 128, 128, 64, 64, 3, 1, 1), (8, 64, 64, 128, 128, 3, 1, 1), (8, 256, 256, 128, 32, 3, 1, 1),
                                  (8, 128, 128, 64, 128, 3, 2, 1), (8, 256, 256, 64, 128, 4, 2, 1), (8, 16, 16, 512, 512, 3, 1, 1)])
def test_uniform_loops_equal_generic_loops_at_full_size(K, case):
    """BASELINE-size layers (8 x 512 x 512 network): the uniform-tap igemm loop and the row-uniform wgrad gather against
    the generic loops of the same kernels.  Forward and dgrad accumulate in the same order -> bit-identical; wgrad adds its
    split partial sums atomically (order varies) -> 1e-5."""
    n, h, w, ci, co, k, s, p = case
    g = torch.Generator(device="cuda").manual_seed(ci + co)
    d = K.conv_desc(n, h, w, ci, co, k, s, p)
    x = torch.randn(n, h, w, ci, device="cuda", generator=g)
    wt = torch.randn(co, k, k, ci, device="cuda", generator=g) / math.sqrt(ci * k * k)
    dy = torch.randn(n, d.ho, d.wo, co, device="cuda", generator=g)
    wtp = torch.empty(ci * k * k * co, device="cuda")
    K.pack_dgrad_weights(d, wt, wtp)
    R = K.bn_replicas()
    outs = {}
    try:
        for mode in (1, 0):
            K.set_generic_gather(mode)
            y = torch.empty(n, d.ho, d.wo, co, device="cuda")
            stats = torch.zeros(R * 2 * co, dtype=torch.float64, device="cuda")
            K.conv2d_fwd_bnstats(d, x, wt, None, y, stats)
            dx = torch.empty_like(x)
            K.conv2d_dgrad(d, dy, wtp, dx)
            dw = torch.empty_like(wt)
            K.conv2d_wgrad(d, x, dy, dw, False)
            outs[mode] = (y, stats.view(R, -1).sum(0), dx, dw)
    finally:
        K.set_generic_gather(-1)
    (y1, s1, dx1, dw1), (y0, s0, dx0, dw0) = outs[1], outs[0]
    assert torch.equal(y0, y1), "fwd: uniform-tap loop differs from the generic loop"
    assert torch.equal(dx0, dx1), "dgrad: uniform-tap loop differs from the generic loop"
    assert torch.allclose(s0, s1, rtol=1e-12, atol=1e-9)
    assert (dw0 - dw1).abs().max().item() <= 1e-5 * dw1.abs().max().item()
    # linearity of the forward in its input (a size-independent property)
    y2 = torch.empty_like(y0)
    K.conv2d_fwd(d, 2.0 * x, wt, None, y2)
    assert torch.equal(y2, 2.0 * y0)


@pytest.mark.parametrize("n,h,w,c1,c2", [(8, 64, 64, 64, 64), (4, 64, 64, 128, 64), (8, 128, 128, 32, 64), (2, 96, 128, 64, 192)])
def test_dgrad_with_bn_backward_reductions(K, n, h, w, c1, c2):
    """udaseg_conv2d_dgrad_bnreduce == udaseg_conv2d_dgrad followed by udaseg_bn_bwd_reduce(dz = dx, z = NULL, y = prev_y):
    dx bit for bit, the two per-channel sums to fp32 block-partial rounding; and the geometry query says no where a tile
    would be ragged, K-sliced or strided."""
    g = torch.Generator().manual_seed(c1 + c2 + h)
    d = K.conv_desc(n, h, w, c1, c2, 3, 1, 1)
    assert K.conv2d_dgrad_bnreduce_ok(d)
    dy = torch.randn(n, h, w, c2, generator=g).cuda()
    wt = (torch.randn(c1, 3, 3, c2, generator=g) / math.sqrt(9 * c2)).cuda()          # already [ci][taps][co]
    prev_y = torch.randn(n, h, w, c1, generator=g).cuda()
    mean, rstd = torch.randn(c1, generator=g).cuda() * 0.1, (torch.rand(c1, generator=g) + 0.5).cuda()
    gamma, beta = (torch.rand(c1, generator=g) + 0.5).cuda(), (torch.randn(c1, generator=g) * 0.3).cuda()
    R = K.bn_replicas()
    dx_ref = torch.empty(n, h, w, c1, device="cuda")
    K.conv2d_dgrad(d, dy, wt, dx_ref)
    bs_ref = torch.zeros(R * 2 * c1, dtype=torch.float64, device="cuda")
    K.bn_bwd_reduce(dx_ref, None, prev_y, mean, rstd, bs_ref, 1, 0.0, gamma=gamma, beta=beta)
    dx = torch.full_like(dx_ref, float("nan"))
    bs = torch.zeros_like(bs_ref)
    K.conv2d_dgrad_bnreduce(d, dy, wt, dx, prev_y, mean, rstd, gamma, beta, 1, 0.0, bs)
    assert torch.equal(dx, dx_ref)
    s, s_ref = bs.view(R, 2, c1).sum(0), bs_ref.view(R, 2, c1).sum(0)
    scale = s_ref.abs().max(dim=1, keepdim=True).values
    assert ((s - s_ref).abs() / scale).max().item() < 1e-5
    # torch restatement of the two sums
    t = prev_y * (gamma * rstd) + (beta - mean * gamma * rstd)
    gm = dx_ref * (t > 0)
    want = torch.stack([gm.double().sum((0, 1, 2)), (gm.double() * ((prev_y - mean) * rstd).double()).sum((0, 1, 2))])
    assert ((s - want).abs() / want.abs().max(dim=1, keepdim=True).values).max().item() < 1e-4
    for bad in (K.conv_desc(8, 63, 65, 64, 64, 3, 1, 1),       # ragged tiles
                K.conv_desc(8, 128, 128, 64, 128, 3, 2, 1),    # strided: parity classes
                K.conv_desc(1, 16, 16, 512, 512, 3, 1, 1),     # K-sliced launch (atomic epilogue)
                K.conv_desc(8, 128, 128, 16, 16, 3, 1, 1)):    # small-channel kernel
        assert not K.conv2d_dgrad_bnreduce_ok(bad)
    # an unsupported geometry with well-formed operands: the LIBRARY says no (error code -> RuntimeError) ...
    dk = K.conv_desc(1, 16, 16, 512, 512, 3, 1, 1)
    z512 = lambda *shape: torch.zeros(*shape, device="cuda")
    with pytest.raises(RuntimeError, match="cannot carry"):
        K.conv2d_dgrad_bnreduce(dk, z512(1, 16, 16, 512), z512(512, 3, 3, 512), z512(1, 16, 16, 512), z512(1, 16, 16, 512), z512(512),
                                z512(512), z512(512), z512(512), 1, 0.0, torch.zeros(R * 2 * 512, dtype=torch.float64, device="cuda"))
    # ... and operands that do not fit the descriptor never reach it (the binding's operand table, _operands.py)
    with pytest.raises(ValueError, match="too short"):
        K.conv2d_dgrad_bnreduce(dk, dy, wt, dx, prev_y, mean, rstd, gamma, beta, 1, 0.0, bs)


def test_host_side_helpers(K):
    """udaseg_memset_async / udaseg_add_i64 / udaseg_stream_wait: what torch.zeros, `num_batches_tracked += 1` and
    `stream.wait_event(Event().record(...))` do for the reference's eager step."""
    z = K.zeros((3, 1000, 7), torch.float64, torch.device("cuda", 0))
    assert z.dtype == torch.float64 and z.shape == (3, 1000, 7) and not z.any()
    base = torch.full((257,), float("nan"), device="cuda")
    zl = K.zeros_like(base)
    assert zl.shape == base.shape and not zl.any() and torch.isnan(base).all()
    cnt = torch.arange(37, device="cuda", dtype=torch.int64) * 1000003
    K.check(K.ops.udaseg_add_i64(cnt, cnt.numel(), 5, None), "add_i64")
    assert torch.equal(cnt.cpu(), torch.arange(37, dtype=torch.int64) * 1000003 + 5)
    # ordering: a side stream waits for a long fill on the main stream before it reads the buffer
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    big = torch.empty(64 << 20, device="cuda")
    out = torch.empty(1, device="cuda")
    for v in (1.0, 2.0, 3.0):
        big.fill_(v)                                             # main stream
        K.stream_wait(side.cuda_stream, main.cuda_stream)
        with torch.cuda.stream(side):
            out.copy_(big[-1:])
        K.stream_wait(main.cuda_stream, side.cuda_stream)        # the next fill must not overtake the read
        torch.cuda.synchronize()
        assert out.item() == v
