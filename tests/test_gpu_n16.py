"""GPU parity of the sixteen-wide matrix tile (csrc/conv_n16_f32x3.hip: v_mfma_f32_16x16x32_bf16, K = two taps of a 16-channel
chunk) for the 3x3 layers of smp.Unet's decoder tail that produce exactly 16 channels.  Reference = torch's CPU float64 convolution
on the same fp32 operands; graded next to the 32-row split kernel (conv3x3_f32x3_kernel) on the same launch, which is itself graded
next to the fp32-MFMA kernel in tests/test_gpu_f32x3.py: norm-wise within 1.25 x + 2^-24, worst element <= 3e-6 of the largest
output.  Forward (plain / unwritten BatchNorm activation as input, BatchNorm statistics), data gradient (BatchNorm-backward sums of
the producing layer), ragged tiles, every supported gathered-channel count, the packing against the oracle's split bit for bit."""
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


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to("cuda", f32)


def nchw(t):
    return t.detach().cpu().permute(0, 3, 1, 2).contiguous()


def err(got, ref64):
    assert got.shape == ref64.shape and torch.isfinite(got).all()
    return ((got.double() - ref64).abs().max() / ref64.abs().max().clamp_min(1e-300)).item()


def err2(got, ref64):
    return ((got.double() - ref64).norm() / ref64.norm()).item()


def pack_n16(K, wt):
    """wt [co][ci][3][3] -> (forward packing if co == 16, data-gradient packing if ci == 16)."""
    co, ci = wt.shape[:2]
    w32 = wt.permute(0, 2, 3, 1).contiguous().cuda()
    wt32 = wt.permute(1, 2, 3, 0).contiguous().cuda()
    rows, off, out = [], 0, {}
    if co == 16:
        n = K.n16_frag_elems(ci)
        rows.append([4, 0, off, 16, ci, ci, 0, 0])
        out["fwd"] = (off, n)
        off += n
    if ci == 16:
        n = K.n16_frag_elems(co)
        rows.append([5, 0, off, 16, co, co, 0, 0])
        out["bwd"] = (off, n)
        off += n
    packed = torch.full((off,), float("nan"), device="cuda", dtype=bf)
    K.pack_up_batched(w32, wt32, packed, torch.tensor(rows, dtype=torch.int32, device="cuda"))
    assert torch.isfinite(packed.float()).all()
    return {k: packed[o:o + n] for k, (o, n) in out.items()}


def test_n16_packing_matches_the_oracle_bit_for_bit(K):
    from oracle.f32x3_ref import split3, up_negated_groups
    g = torch.Generator().manual_seed(2)
    wt = torch.randn(16, 24, 3, 3, generator=g) * torch.exp2(torch.randint(-10, 10, (16, 24, 3, 3), generator=g).float())
    P = pack_n16(K, wt)
    nk = 2
    planes = P["fwd"].view(3, nk * 5, 64, 8).float().cpu().numpy()
    want = np.zeros((3, nk * 5, 64, 8), dtype=np.float32)
    neg = up_negated_groups(5 * nk)
    w = wt.numpy()
    for G in range(nk * 5):
        kk, j = divmod(G, 5)
        for l in range(64):
            n_, gsl = l & 15, l >> 4
            tap, k0 = 2 * j + (gsl >> 1), kk * 16 + 8 * (gsl & 1)
            if tap < 9 and k0 < 24:
                v = w[n_, k0:k0 + 8, tap // 3, tap % 3]
                t = split3(-v if G in neg else v)
                for pl in range(3):
                    want[pl, G, l] = t[pl]
    assert np.array_equal(planes, want)


FWD_CASES = [(2, 16, 32, 16), (1, 24, 40, 16), (1, 20, 72, 24), (2, 9, 33, 32), (1, 40, 40, 8), (8, 128, 128, 16), (1, 512, 512, 16)]


@pytest.mark.parametrize("case", FWD_CASES, ids=[("n%d_%dx%d_ci%d" % c) for c in FWD_CASES])
@pytest.mark.parametrize("lazy", [False, True], ids=["plain", "unwritten_bn_input"])
def test_n16_forward_fp32_grade(K, case, lazy):
    from test_gpu_f32x3 import pack3
    n, h, w, ci = case
    co = 16
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    d = K.conv_desc(n, h, w, ci, co, 3, 1, 1)
    assert K.conv_n16_ok(d)
    P = pack_n16(K, wt)
    R = K.bn_replicas()
    xd = nhwc(x)
    kw = {}
    xin = x.double()
    if lazy:
        sc, sh = (torch.rand(ci, generator=g) + 0.5), (torch.randn(ci, generator=g) * 0.5)
        kw = dict(in_scale=sc.cuda(), in_shift=sh.cuda(), in_act=1, in_slope=0.0)
        z = torch.addcmul(sh.view(1, -1, 1, 1).double(), x.double(), sc.view(1, -1, 1, 1).double()).float()     # one rounding, like the fma
        xin = torch.relu(z).double()
    y_ref = F.conv2d(xin, wt.double(), padding=1)
    y = torch.full((n, h, w, co), float("nan"), device="cuda", dtype=f32)
    st = torch.zeros(R * 2 * co, dtype=f64, device="cuda")
    K.conv2d_fwd_n16(d, xd, P["fwd"], y, stats=st, **kw)
    # the 32-row split kernel on the same launch
    wf, _, _, _ = pack3(K, wt)
    y32 = torch.empty_like(y)
    K.conv2d_fwd_frag(d, xd, None, wf, None, y32, **kw)
    e2, n2, em = err2(nchw(y), y_ref), err2(nchw(y32), y_ref), err(nchw(y), y_ref)
    print(f"forward: l2 sixteen-wide {e2:.3e} 32-row {n2:.3e} | worst element {em:.3e}")
    assert e2 <= 1.25 * n2 + 2.0 ** -24 and em <= 3e-6
    tot = st.view(R, 2, co).sum(0).cpu()
    yd = y_ref.permute(0, 2, 3, 1).reshape(-1, co)
    assert (tot[0] - yd.sum(0)).abs().max().item() <= 1e-5 * yd.abs().sum(0).max().item(), "fused sum"
    assert ((tot[1] - (yd * yd).sum(0)).abs().max() / (yd * yd).sum(0).max()).item() <= 1e-5, "fused sum of squares"
    # the statistics are those of the tensor the kernel stored (BatchNorm's backward relies on sum(y_hat) == 0 over the stored y)
    own = y.double().reshape(-1, y.shape[-1])
    e_own = ((tot[0] - own.sum(0).cpu()).abs().max() / own.abs().sum(0).max().cpu()).item()
    print(f"statistics against the stored tensor's own sums: {e_own:.2e}")
    assert e_own <= 2e-7


BWD_CASES = [(2, 16, 32, 16), (1, 24, 40, 24), (1, 20, 72, 16), (2, 9, 33, 24), (8, 128, 128, 24), (1, 512, 512, 16)]


@pytest.mark.parametrize("case", BWD_CASES, ids=[("n%d_%dx%d_co%d" % c) for c in BWD_CASES])
@pytest.mark.parametrize("bnb", [False, True], ids=["plain", "bn_backward_sums"])
def test_n16_data_gradient_fp32_grade(K, case, bnb):
    from test_gpu_f32x3 import pack3
    n, h, w, co = case
    ci = 16
    g = torch.Generator().manual_seed(sum(case) + 7)
    wt = torch.randn(co, ci, 3, 3, generator=g) / math.sqrt(ci * 9)
    dy = torch.randn(n, co, h, w, generator=g)
    x64 = torch.zeros(n, ci, h, w, dtype=f64, requires_grad=True)
    F.conv2d(x64, wt.double(), padding=1).backward(dy.double())
    d = K.conv_desc(n, h, w, ci, co, 3, 1, 1)
    assert K.conv_n16_ok(d, dgrad=True)
    P = pack_n16(K, wt)
    R = K.bn_replicas()
    dyd = nhwc(dy)
    dx = torch.full((n, h, w, ci), float("nan"), device="cuda", dtype=f32)
    bn = None
    if bnb:
        prev_y = torch.randn(n, h, w, ci, generator=g)
        mean, var = prev_y.reshape(-1, ci).mean(0), prev_y.reshape(-1, ci).var(0, unbiased=False)
        rstd = (var + 1e-5).rsqrt()
        gamma, beta = torch.rand(ci, generator=g) + 0.5, torch.randn(ci, generator=g) * 0.3
        bs = torch.zeros(R * 2 * ci, dtype=f64, device="cuda")
        bn = (prev_y.cuda(), mean.cuda(), rstd.cuda(), gamma.cuda(), beta.cuda(), 1, 0.0, bs)
    K.conv2d_dgrad_n16(d, dyd, P["bwd"], dx, bn=bn)
    _, wfd, _, _ = pack3(K, wt)
    dx32 = torch.empty_like(dx)
    K.conv2d_dgrad_frag(d, dyd, wfd, dx32)
    e2, n2, em = err2(nchw(dx), x64.grad), err2(nchw(dx32), x64.grad), err(nchw(dx), x64.grad)
    print(f"data gradient: l2 sixteen-wide {e2:.3e} 32-row {n2:.3e} | worst element {em:.3e}")
    assert e2 <= 1.25 * n2 + 2.0 ** -24 and em <= 3e-6
    if bnb:
        gz = dx.cpu().double().reshape(-1, ci)
        yy = prev_y.double().reshape(-1, ci)
        scd = (gamma * rstd).double()
        arg = yy * scd + (beta.double() - mean.double() * scd)
        gg = gz * (arg > 0).double()
        s1, s2 = gg.sum(0), (gg * (yy - mean.double()) * rstd.double()).sum(0)
        tot = bs.view(R, 2, ci).sum(0).cpu()
        assert ((tot[0] - s1).abs().max() / s1.abs().max()).item() <= 1e-5
        assert ((tot[1] - s2).abs().max() / s2.abs().max()).item() <= 1e-5


STEM_CASES = [(2, 64, 64), (1, 48, 80), (8, 256, 256), (1, 34, 38), (1, 2, 2), (2, 512, 512), (8, 512, 512), (5, 1024, 768)]


@pytest.mark.parametrize("case", STEM_CASES, ids=[("n%d_%dx%d" % c) for c in STEM_CASES])
def test_stem_forward_fp32_grade(K, case):
    """The 7x7 / stride 2 / pad 3 stem (3 + 1 padding channels -> 64) on csrc/conv_stem_f32x3.hip against float64, next to the shared
    implicit-GEMM source on the fp32 matrix pipe; BatchNorm statistics; ragged tiles; the zero-weight columns that pad a kernel row's
    28 values to K = 32 read the neighbouring pixel and must contribute nothing (the image's 4th channel is NOT zero here)."""
    n, h, w = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, 4, h, w, generator=g) if n != 8 else torch.rand(n, 4, h, w, generator=g)     # n == 8: images in [0, 1)
    wt = torch.randn(64, 4, 7, 7, generator=g) / math.sqrt(4 * 49)
    y_ref = F.conv2d(x.double(), wt.double(), stride=2, padding=3)
    d = K.conv_desc(n, h, w, 4, 64, 7, 2, 3)
    assert K.conv_stem_ok(d)
    w32 = wt.permute(0, 2, 3, 1).contiguous().cuda()
    packed = torch.full((K.STEM_FRAG_ELEMS,), float("nan"), device="cuda", dtype=bf)
    K.pack_up_batched(w32, None, packed, torch.tensor([[8, 0, 0, 64, 4, 4, 0, 0]], dtype=torch.int32, device="cuda"))
    assert torch.isfinite(packed.float()).all()
    R = K.bn_replicas()
    xd = nhwc(x)
    y = torch.full((n, h // 2, w // 2, 64), float("nan"), device="cuda", dtype=f32)
    st = torch.zeros(R * 2 * 64, dtype=f64, device="cuda")
    K.conv2d_fwd_stem(d, xd, packed, y, stats=st)
    y_nat = torch.empty_like(y)
    K.set_f32_split(0)
    try:
        K.conv2d_fwd(d, xd, w32, None, y_nat, 0, 0.0, False)
    finally:
        K.set_f32_split(-1)
    e2, n2, em = err2(nchw(y), y_ref), err2(nchw(y_nat), y_ref), err(nchw(y), y_ref)
    print(f"stem forward: l2 {e2:.3e} fp32-MFMA implicit GEMM {n2:.3e} | worst element {em:.3e}")
    assert e2 <= 1.25 * n2 + 2.0 ** -24 and em <= 3e-6
    tot = st.view(R, 2, 64).sum(0).cpu()
    yd = y_ref.permute(0, 2, 3, 1).reshape(-1, 64)
    assert (tot[0] - yd.sum(0)).abs().max().item() <= 1e-5 * yd.abs().sum(0).max().item(), "fused sum"
    assert ((tot[1] - (yd * yd).sum(0)).abs().max() / (yd * yd).sum(0).max()).item() <= 1e-5, "fused sum of squares"
    # the statistics are those of the tensor the kernel stored (BatchNorm's backward relies on sum(y_hat) == 0 over the stored y)
    own = y.double().reshape(-1, y.shape[-1])
    e_own = ((tot[0] - own.sum(0).cpu()).abs().max() / own.abs().sum(0).max().cpu()).item()
    print(f"statistics against the stored tensor's own sums: {e_own:.2e}")
    assert e_own <= 2e-7
