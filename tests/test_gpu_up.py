"""GPU parity of the decoder's up-sampled input convolved as four 2x2 phase convolutions (csrc/conv_up_f32x3.hip).

Reference = torch's CPU float64 evaluation of the reference's own op sequence, conv2d(cat([interpolate(a, 2, 'nearest'), skip]),
padding=1) and its autograd (smp DecoderBlock; src/models/train.py:341,343), on the same fp32 operands.  Graded next to the nine-tap
kernel that gathers the same virtual input (udaseg_conv2d_fwd_f32x3 with up_ca): norm-wise within 1.5 x its error + 2^-24 and
<= 1e-5 outright (bit equality with the materialised path is NOT expected -- the pre-summed weights round
once more, and the sums run in another order).  The packings are compared with oracle/f32x3_ref.py bit for bit.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

f32, f64, bf = torch.float32, torch.float64, torch.bfloat16


@pytest.fixture(scope="module")
def K():
    from uda_aerial_semantic_segmentation_research_amd import _lib, kernels
    _lib.require_gpu()
    kernels.ensure_workspace(torch.device("cuda", 0))
    return kernels


@pytest.fixture
def force_up(K):
    from uda_aerial_semantic_segmentation_research_amd import _lib
    lib = _lib.load()
    yield lambda c: lib.udaseg_up_f32x3_force_config(c)
    lib.udaseg_up_f32x3_force_config(0)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to("cuda", f32)


def nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2).contiguous()


def err(got, ref64):
    assert got.shape == ref64.shape, (got.shape, ref64.shape)
    assert torch.isfinite(got).all()
    return ((got.double() - ref64).abs().max() / ref64.abs().max().clamp_min(1e-300)).item()


def err2(got, ref64):
    return ((got.double() - ref64).norm() / ref64.norm()).item()


def pack_up(K, wt, ca):
    """wt [co][ci][3][3] fp32, the first ca input channels up-sampled -> dict of packings + the plain sources."""
    co, ci = wt.shape[:2]
    cs = ci - ca
    w32 = wt.permute(0, 2, 3, 1).contiguous().cuda()            # OHWI
    wt32 = wt.permute(1, 2, 3, 0).contiguous().cuda()           # [ci][kh][kw][co]: the dgrad packing
    n_uf, n_ub = 3 * K.frag_elems(co, ca, 4), 3 * K.frag_elems(ca, co, 4)
    n_sf = 3 * K.frag_elems(co, cs, 3) if cs else 0
    n_sb = 3 * K.frag_elems(cs, co, 3) if cs else 0
    packed = torch.full((n_uf + n_ub + n_sf + n_sb,), float("nan"), device="cuda", dtype=bf)
    rows = [[2, 0, 0, co, ca, ci, 0, 0], [3, 0, n_uf, ca, co, co, 0, 0]]
    if cs:
        rows += [[0, ca, n_uf + n_ub, co, cs, ci, 0, 0], [1, ca * 9 * co, n_uf + n_ub + n_sf, cs, co, co, 0, 0]]
    K.pack_up_batched(w32, wt32, packed, torch.tensor(rows, dtype=torch.int32, device="cuda"))
    assert torch.isfinite(packed.float()).all()
    o = [0, n_uf, n_uf + n_ub, n_uf + n_ub + n_sf, n_uf + n_ub + n_sf + n_sb]
    return {"up_fwd": packed[o[0]:o[1]], "up_bwd": packed[o[1]:o[2]], "skip_fwd": packed[o[2]:o[3]] if cs else None,
            "skip_bwd": packed[o[3]:o[4]] if cs else None, "w32": w32, "wt32": wt32}


def test_phase_packings_match_the_oracle_bit_for_bit(K):
    """plane0 + plane1 + plane2 of every fragment == the oracle's fp32 pre-summed weight (sign per group), both packings."""
    from oracle.f32x3_ref import phase_weights, up_negated_groups
    g = torch.Generator().manual_seed(5)
    co, ca, cs = 40, 48, 16
    wt = torch.randn(co, ca + cs, 3, 3, generator=g) * torch.exp2(torch.randint(-12, 12, (co, ca + cs, 3, 3), generator=g).float())
    P = pack_up(K, wt, ca)
    pw = phase_weights(wt[:, :ca].numpy())                      # [py][px][co][ca][u][v]
    lane = np.arange(64)
    # forward: plane[nb][G = 4 chunk + g][j][lane][8]; g = (px, ex) in {(0,0),(0,1),(1,1),(1,2)}; j = (ey, py) in {(0,0),(1,0),(1,1),(2,1)}
    nb, nk = (co + 31) // 32, (ca + 15) // 16
    tot = P["up_fwd"].view(3, -1).float().cpu().double().sum(0).view(nb, 4 * nk, 4, 64, 8).numpy()
    neg = up_negated_groups(4 * nk)
    for b in range(nb):
        for G in range(4 * nk):
            kk, gi = G // 4, G % 4
            px, ex = gi >> 1, (gi + 1) >> 1
            for j in range(4):
                py, u, v = j >> 1, j & 1, ex - px
                want = np.zeros((64, 8))
                for l in lane:
                    n_, k0 = b * 32 + (l & 31), kk * 16 + 8 * (l >> 5)
                    if n_ < co and k0 < ca:
                        want[l] = pw[py, px, n_, k0:k0 + 8, u, v]
                if G in neg:
                    want = -want
                assert np.array_equal(tot[b, G, j], want), ("fwd", b, G, j)
    # data gradient: plane[nb][G = 4 chunk + phase][j = 2 e + jy][lane][8]; n = a channel, k = dy channel
    nb, nk = (ca + 31) // 32, (co + 15) // 16
    tot = P["up_bwd"].view(3, -1).float().cpu().double().sum(0).view(nb, 4 * nk, 4, 64, 8).numpy()
    neg = up_negated_groups(4 * nk)
    for b in range(nb):
        for G in range(4 * nk):
            kk, py, px = G >> 2, (G >> 1) & 1, G & 1
            for j in range(4):
                e, jy = j >> 1, j & 1
                ex = e if px else 1 + e
                ey = (0 if py else 1) + jy
                u, v = 2 - py - ey, 2 - px - ex
                want = np.zeros((64, 8))
                for l in lane:
                    n_, k0 = b * 32 + (l & 31), kk * 16 + 8 * (l >> 5)
                    if n_ < ca and k0 < co:
                        hi = min(k0 + 8, co)
                        want[l, :hi - k0] = pw[py, px, k0:hi, n_, u, v]
                if G in neg:
                    want = -want
                assert np.array_equal(tot[b, G, j], want), ("bwd", b, G, j)


# (n, h, w of the OUTPUT, up-sampled channels, skip channels, produced channels)
CASES = [(2, 16, 64, 64, 32, 64), (1, 20, 72, 32, 32, 32), (2, 32, 64, 32, 0, 16), (1, 8, 64, 512, 256, 256), (2, 32, 32, 128, 64, 64),
         (1, 12, 24, 48, 16, 24), (8, 64, 64, 256, 128, 128), (1, 2, 2, 16, 0, 8), (3, 10, 6, 16, 16, 40), (1, 64, 128, 64, 64, 32)]
CFG_IDS = ["heuristic", "2x32x64ch", "4x32x64ch", "4x32x32ch", "8x32x32ch", "4x16x64ch", "8x16x64ch", "8x16x32ch", "8x32x64ch_dgrad"]


@pytest.mark.parametrize("case", CASES, ids=[("n%d_%dx%d_ca%d_cs%d_co%d" % c) for c in CASES])
@pytest.mark.parametrize("cfg", list(range(9)), ids=CFG_IDS)
def test_up_phase_conv_fwd_dgrad_fp32_grade(K, case, cfg, force_up):
    n, h, w, ca, cs, co = case
    force_up(cfg)
    g = torch.Generator().manual_seed(sum(case))
    a = torch.randn(n, ca, h // 2, w // 2, generator=g)
    skip = torch.randn(n, cs, h, w, generator=g) if cs else None
    ci = ca + cs
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    a64 = a.double().requires_grad_(True)
    s64 = skip.double().requires_grad_(True) if cs else None
    up = F.interpolate(a64, scale_factor=2, mode="nearest")
    y_ref = F.conv2d(torch.cat([up, s64], 1) if cs else up, wt.double(), padding=1)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.double())
    d = K.conv_desc(n, h, w, ci, co, 3, 1, 1)
    assert K.conv_up_ok(d, ca)
    P = pack_up(K, wt, ca)
    R = K.bn_replicas()
    ad, sd = nhwc(a), (nhwc(skip) if cs else None)

    # ---- forward: the skip half by the plain nine-tap kernel, the up-sampled half on top, statistics of the sum
    y = torch.full((n, h, w, co), float("nan"), device="cuda", dtype=f32)
    st = torch.zeros(R * 2 * co, dtype=f64, device="cuda")
    if cs:
        K.conv2d_fwd_frag(K.conv_desc(n, h, w, cs, co, 3, 1, 1), sd, None, P["skip_fwd"], None, y)
    K.conv2d_fwd_up(d, ad, P["up_fwd"], y, accumulate=bool(cs), stats=st)
    # the nine-tap kernel over the same virtual input, for the comparison
    from test_gpu_f32x3 import pack3
    wf9, wfd9, _, _ = pack3(K, wt)
    y9 = torch.empty_like(y)
    if K.conv_frag_ok(d, up_ca=ca, f32=True):
        K.conv2d_fwd_frag(d, ad, sd, wf9, None, y9, up=True)
        e9 = err2(nchw(y9), y_ref.detach())
    else:
        e9 = 4e-7
    e2, em = err2(nchw(y), y_ref.detach()), err(nchw(y), y_ref.detach())
    print(f"forward: l2 phase {e2:.3e} nine-tap {e9:.3e} | worst element {em:.3e}")
    assert e2 <= 1.5 * e9 + 2.0 ** -24 and em <= 1e-5
    tot = st.view(R, 2, co).sum(0).cpu()
    yd = y_ref.detach().permute(0, 2, 3, 1).reshape(-1, co)
    assert (tot[0] - yd.sum(0)).abs().max().item() <= 1e-5 * yd.abs().sum(0).max().item(), "fused sum"
    assert ((tot[1] - (yd * yd).sum(0)).abs().max() / (yd * yd).sum(0).max()).item() <= 1e-5, "fused sum of squares"

    # ---- data gradient of a at a's own resolution (convolution transpose + the up-sampling's 2x2 sum in one pass)
    dyd = nhwc(dy)
    da = torch.full((n, h // 2, w // 2, ca), float("nan"), device="cuda", dtype=f32)
    if co % 8 == 0:
        K.conv2d_dgrad_up(d, dyd, ca, P["up_bwd"], da)
        e2, em = err2(nchw(da), a64.grad), err(nchw(da), a64.grad)
        print(f"data gradient: l2 {e2:.3e} worst element {em:.3e}")
        assert e2 <= 1e-6 and em <= 1e-5
        base = torch.randn(n, ca, h // 2, w // 2, generator=g)
        dacc = nhwc(base)
        K.conv2d_dgrad_up(d, dyd, ca, P["up_bwd"], dacc, accumulate=True)
        assert err(nchw(dacc), a64.grad + base.double()) <= 1e-5
    if cs and cs % 8 == 0:
        dsk = torch.full((n, h, w, cs), float("nan"), device="cuda", dtype=f32)
        K.conv2d_dgrad_frag(K.conv_desc(n, h, w, cs, co, 3, 1, 1), dyd, P["skip_bwd"], dsk)
        assert err(nchw(dsk), s64.grad) <= 3e-6


def test_up_only_write_mode_leaves_nothing_unwritten(K):
    """No skip source: the kernel overwrites y (NaN-filled here) at every pixel and channel, ragged tiles included."""
    n, h, w, ca, co = 1, 36, 44, 32, 16
    g = torch.Generator().manual_seed(1)
    a = torch.randn(n, ca, h // 2, w // 2, generator=g)
    wt = torch.randn(co, ca, 3, 3, generator=g) / math.sqrt(ca * 9)
    P = pack_up(K, wt, ca)
    y = torch.full((n, h, w, co), float("nan"), device="cuda", dtype=f32)
    K.conv2d_fwd_up(K.conv_desc(n, h, w, ca, co, 3, 1, 1), nhwc(a), P["up_fwd"], y)
    ref = F.conv2d(F.interpolate(a.double(), scale_factor=2, mode="nearest"), wt.double(), padding=1)
    assert err(nchw(y), ref) <= 1e-5


WG_CASES = [(2, 16, 64, 64, 64, 64), (2, 32, 32, 128, 64, 64), (1, 20, 72, 64, 32, 32), (8, 64, 64, 256, 128, 128),
            (1, 8, 64, 512, 256, 256), (2, 24, 40, 64, 0, 64), (3, 34, 70, 64, 64, 96), (2, 34, 38, 64, 64, 64), (8, 128, 128, 128, 64, 64)]


@pytest.mark.parametrize("case", WG_CASES, ids=[("n%d_%dx%d_ca%d_cs%d_co%d" % c) for c in WG_CASES])
@pytest.mark.parametrize("blocks", [0, 16, 600], ids=["default_blocks", "16_blocks", "600_blocks"])
def test_up_phase_wgrad_fp32_grade(K, case, blocks):
    """dW of conv3x3(cat([nearest_x2(a), skip])): the up-sampled half as 16 phase-tap correlations at a's resolution
    (conv_wgrad_up_kernel), the skip half as a channel slice of the halo-resident kernel -- against the float64 autograd of the
    reference's op sequence, next to the nine-tap halo kernel over the same virtual input; accumulation onto an existing gradient;
    any split of the tile sequence over blocks (atomics) gives the same sums."""
    from uda_aerial_semantic_segmentation_research_amd import _lib
    n, h, w, ca, cs, co = case
    g = torch.Generator().manual_seed(sum(case))
    a = torch.randn(n, ca, h // 2, w // 2, generator=g)
    skip = torch.randn(n, cs, h, w, generator=g) if cs else None
    dy = torch.randn(n, co, h, w, generator=g)
    up = F.interpolate(a, scale_factor=2, mode="nearest")
    xin = (torch.cat([up, skip], 1) if cs else up).double()
    wt = torch.zeros(co, ca + cs, 3, 3, dtype=f64, requires_grad=True)
    F.conv2d(xin, wt, padding=1).backward(dy.double())
    ref = wt.grad.permute(0, 2, 3, 1).contiguous()              # OHWI
    d = K.conv_desc(n, h, w, ca + cs, co, 3, 1, 1)
    assert K.conv2d_wgrad_up_ok(d, ca)
    base = torch.randn(co, 3, 3, ca + cs, generator=g)
    dw = base.clone().cuda()
    lib = _lib.load()
    lib.udaseg_wgrad_up_set_blocks(blocks)
    try:
        K.conv2d_wgrad_up(d, nhwc(a), nhwc(dy), dw)
    finally:
        lib.udaseg_wgrad_up_set_blocks(0)
    if cs:
        K.conv2d_wgrad_halo_slice(K.conv_desc(n, h, w, cs, co, 3, 1, 1), nhwc(skip), nhwc(dy), dw, ca)
    got = dw.cpu() - base
    e2, em = err2(got, ref), err(got, ref)
    e9 = 5e-7
    if cs and K.conv2d_wgrad_halo_ok(d, ca, f32=True):
        dw9 = torch.zeros(co, 3, 3, ca + cs, device="cuda")
        K.conv2d_wgrad_halo(d, nhwc(a), nhwc(skip), nhwc(dy), dw9, up=True)
        e9 = err2(dw9.cpu(), ref)
    print(f"weight gradient: l2 phase {e2:.3e} nine-tap {e9:.3e} | worst element {em:.3e}")
    assert e2 <= 1.6 * e9 + 2.0 ** -23 and em <= 1e-5
    # the up-sampled half alone, tap by tap: every one of the nine taps got its four phase taps
    for ky in range(3):
        for kx in range(3):
            assert err2(got[:, ky, kx, :ca], ref[:, ky, kx, :ca]) <= 2e-6, (ky, kx)


@pytest.mark.parametrize("n,h,w,ca,co,act,slope", [(2, 32, 64, 64, 64, 1, 0.0), (1, 20, 36, 32, 16, 1, 0.2), (8, 64, 64, 128, 64, 1, 0.0),
                                                    (2, 32, 32, 256, 128, 1, 0.0)])
def test_up_phase_dgrad_with_bn_backward_reductions(K, n, h, w, ca, co, act, slope):
    """The phase data gradient's epilogue also makes the two BatchNorm-backward sums of the layer that produced `a` (its only consumer
    is this convolution, and da is complete here -- no 2x2 sum-pool follows): the gradient is untouched, the sums equal the
    definition in f64 on the stored gradient."""
    g = torch.Generator().manual_seed(n * 1000 + h + ca)
    R = K.bn_replicas()
    wt = torch.randn(co, ca, 3, 3, generator=g) / math.sqrt(ca * 9)
    dy = torch.randn(n, co, h, w, generator=g)
    prev_y = torch.randn(n, h // 2, w // 2, ca, generator=g)
    mean, var = prev_y.reshape(-1, ca).mean(0), prev_y.reshape(-1, ca).var(0, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    gamma, beta = torch.rand(ca, generator=g) + 0.5, torch.randn(ca, generator=g) * 0.3
    d = K.conv_desc(n, h, w, ca, co, 3, 1, 1)
    P = pack_up(K, wt, ca)
    dyd = nhwc(dy)
    da = torch.full((n, h // 2, w // 2, ca), float("nan"), device="cuda", dtype=f32)
    bs = torch.zeros(R * 2 * ca, dtype=f64, device="cuda")
    K.conv2d_dgrad_up(d, dyd, ca, P["up_bwd"], da, bn=(prev_y.cuda(), mean.cuda(), rstd.cuda(), gamma.cuda(), beta.cuda(), act, slope, bs))
    plain = torch.empty_like(da)
    K.conv2d_dgrad_up(d, dyd, ca, P["up_bwd"], plain)
    assert torch.equal(da, plain)
    gz = da.cpu().double().reshape(-1, ca)
    yy = prev_y.double().reshape(-1, ca)
    sc = (gamma * rstd).double()
    arg = yy * sc + (beta.double() - mean.double() * sc)
    mask = torch.where(arg > 0, torch.ones_like(arg), torch.full_like(arg, slope))
    gg = gz * mask
    s1, s2 = gg.sum(0), (gg * (yy - mean.double()) * rstd.double()).sum(0)
    tot = bs.view(R, 2, ca).sum(0).cpu()
    assert ((tot[0] - s1).abs().max() / s1.abs().max()).item() <= 1e-5
    assert ((tot[1] - s2).abs().max() / s2.abs().max()).item() <= 1e-5
