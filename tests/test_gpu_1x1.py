"""GPU parity of the hand-written 1x1 / stride-1 bf16 convolutions on r50's small-GEMM shapes (csrc/conv_halo_bf16.hip: streaming
and small-GEMM kernels, forward / data gradient with fused BatchNorm statistics; csrc/conv_wgrad.hip: weight gradient) against
fp32 torch on the bf16-rounded operands (the bar of tests/test_gpu_bf16.py: 2^-7 norm-wise)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


@pytest.fixture(scope="module")
def K():
    from uda_aerial_semantic_segmentation_research_amd import _lib, kernels
    _lib.require_gpu()
    kernels.ensure_workspace(torch.device("cuda", 0))
    return kernels


def rel(got, ref):
    return ((got.double() - ref.double()).norm() / ref.double().norm()).item()


def pack(K, wt):
    """wt: [co][ci] bf16-representable fp32 -> (forward fragments, data-gradient fragments) of the 1x1 convolution."""
    co, ci = wt.shape
    w16 = wt.contiguous().to("cuda", bf)                 # OHWI with a 1x1 window
    wt16 = wt.t().contiguous().to("cuda", bf)            # [ci][1][1][co]
    nf, nd = K.frag_elems(co, ci, 1), K.frag_elems(ci, co, 1)
    packed = torch.full((nf + nd,), float("nan"), device="cuda", dtype=bf)
    table = torch.tensor([[0, 0, 0, co, ci, 1], [1, 0, nf, ci, co, 1]], dtype=torch.int32, device="cuda")
    K.pack_frag_batched(w16, wt16, packed, table)
    return packed[:nf], packed[nf:]


# r50's bottleneck projections as GEMMs: [pixels][ci] x [ci][co] (reference smp.Unet("resnet50"), src/test_system.py:90-95, driven
# src/models/train.py:341,343) -- the shapes round 3 had handed to a vendor library, on the repo's own kernels
@pytest.mark.parametrize("n,h,w,ci,co", [(8, 48, 48, 256, 1024), (8, 48, 48, 1024, 256), (2, 24, 24, 512, 2048), (1, 7, 9, 64, 128),
                                         (8, 96, 96, 128, 512), (3, 5, 5, 2048, 512), (1, 1, 1, 64, 64)])
def test_conv1x1_fwd_dgrad_wgrad(K, n, h, w, ci, co):
    g = torch.Generator().manual_seed(n + h + ci + co)
    x = torch.randn(n, h, w, ci, generator=g).to(bf)
    wt = (torch.randn(co, ci, generator=g) / math.sqrt(ci)).to(bf)
    dy = torch.randn(n, h, w, co, generator=g).to(bf)
    x32, w32, dy32 = x.float(), wt.float(), dy.float()
    xd, dyd = x.cuda(), dy.cuda()
    d = K.conv_desc(n, h, w, ci, co, 1, 1, 0)
    assert K.conv_frag_ok(d) and K.conv_frag_ok(d, dgrad=True)
    wf, wfd = pack(K, w32)
    R = K.bn_replicas()
    y = torch.full((n, h, w, co), float("nan"), device="cuda", dtype=bf)
    st = torch.zeros(R * 2 * co, dtype=torch.float64, device="cuda")
    K.conv2d_fwd_frag(d, xd, None, wf, None, y, stats=st)
    y_ref = x32 @ w32.t()
    assert rel(y.float().cpu(), y_ref) <= 2.0 ** -7
    tot = st.view(R, 2, co).sum(0).cpu()                  # BatchNorm statistics from the fp32 accumulators, same launch
    yd = y_ref.double().reshape(-1, co)
    assert (tot[0] - yd.sum(0)).abs().max().item() <= 1e-4 * yd.abs().sum(0).max().item()
    assert rel(tot[1], (yd * yd).sum(0)) <= 1e-3
    dx = torch.full((n, h, w, ci), float("nan"), device="cuda", dtype=bf)
    K.conv2d_dgrad_frag(d, dyd, wfd, dx)
    ref_dx = dy32 @ w32
    assert rel(dx.float().cpu(), ref_dx) <= 2.0 ** -7
    base = torch.randn(n, h, w, ci, generator=g).to(bf)
    dxa = base.cuda()
    K.conv2d_dgrad_frag(d, dyd, wfd, dxa, accumulate=True)
    assert rel(dxa.float().cpu(), ref_dx + base.float()) <= 2.0 ** -7
    gbase = torch.randn(co, ci, generator=g)
    dw = gbase.cuda().view(co, 1, 1, ci)
    K.conv2d_wgrad(d, xd, dyd, dw, accumulate=True)
    ref_dw = dy32.reshape(-1, co).t() @ x32.reshape(-1, ci)
    assert rel(dw.cpu().view(co, ci) - gbase, ref_dw) <= 1e-3          # fp32 output of bf16 operands: only the summation order differs


@pytest.mark.parametrize("n,h,w,c", [(8, 48, 48, 1024), (2, 7, 9, 64), (8, 96, 96, 512), (1, 5, 5, 2048), (3, 11, 13, 40)])
def test_bn_stats_bf16(K, n, h, w, c):
    """Per-channel sum / sum of squares of a bf16 tensor (the stand-alone statistics pass) against float64; the fp32 entry point
    refuses a bf16 tensor at the binding instead of reading it as fp32."""
    g = torch.Generator().manual_seed(c)
    y = (torch.randn(n, h, w, c, generator=g) * 1.7 + 0.3).to(bf)
    R = K.bn_replicas()
    sums = torch.zeros(R * 2 * c, dtype=torch.float64, device="cuda")
    K.bn_stats(y.cuda(), sums)
    tot = sums.view(R, 2, c).sum(0).cpu()
    yd = y.double().reshape(-1, c)
    assert ((tot[0] - yd.sum(0)).abs().max() / yd.abs().sum(0).max()).item() <= 1e-12
    assert ((tot[1] - (yd * yd).sum(0)).abs().max() / (yd * yd).sum(0).max()).item() <= 1e-12
    with pytest.raises(ValueError):
        K.bn_stats(y.cuda().to(torch.float16), sums)
