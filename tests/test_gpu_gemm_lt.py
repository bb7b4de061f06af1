"""GPU parity of the library-GEMM path of the 1x1 / stride-1 bf16 convolutions (csrc/gemm_lt.hip, hipBLASLt) against fp32 torch on
the bf16-rounded operands (the bar of tests/test_gpu_bf16.py: 2^-7 norm-wise), and against the hand-written kernels."""
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


@pytest.mark.parametrize("n,h,w,ci,co", [(8, 48, 48, 256, 1024), (8, 48, 48, 1024, 256), (2, 24, 24, 512, 2048), (1, 7, 9, 64, 128),
                                         (8, 96, 96, 128, 512), (3, 5, 5, 2048, 512), (1, 1, 1, 64, 64)])
def test_gemm_1x1_fwd_dgrad_wgrad(K, n, h, w, ci, co):
    g = torch.Generator().manual_seed(n + h + ci + co)
    x = torch.randn(n, h, w, ci, generator=g).to(bf)
    wt = (torch.randn(co, ci, generator=g) / math.sqrt(ci)).to(bf)
    dy = torch.randn(n, h, w, co, generator=g).to(bf)
    x32, w32, dy32 = x.float(), wt.float(), dy.float()
    xd, wd, dyd = x.cuda(), wt.cuda(), dy.cuda()
    y = torch.full((n, h, w, co), float("nan"), device="cuda", dtype=bf)
    K.gemm_1x1(0, xd, wd, y)
    assert rel(y.float().cpu(), x32 @ w32.t()) <= 2.0 ** -7
    dx = torch.full((n, h, w, ci), float("nan"), device="cuda", dtype=bf)
    K.gemm_1x1(1, dyd, wd, dx)
    ref_dx = dy32 @ w32
    assert rel(dx.float().cpu(), ref_dx) <= 2.0 ** -7
    base = torch.randn(n, h, w, ci, generator=g).to(bf)
    dxa = base.cuda()
    K.gemm_1x1(1, dyd, wd, dxa, accumulate=True)
    assert rel(dxa.float().cpu(), ref_dx + base.float()) <= 2.0 ** -7
    gbase = torch.randn(co, ci, generator=g)
    dw = gbase.cuda()
    K.gemm_1x1(2, xd, dyd, dw, accumulate=True)
    ref_dw = dy32.reshape(-1, co).t() @ x32.reshape(-1, ci)
    assert rel(dw.cpu() - gbase, ref_dw) <= 1e-3          # fp32 output of bf16 operands: only the summation order differs
    d = K.conv_desc(n, h, w, ci, co, 1, 1, 0)
    assert K.gemm_1x1_preferred(d) == (n * h * w <= 73728)


def test_gemm_1x1_refuses_bad_arguments(K):
    from uda_aerial_semantic_segmentation_research_amd import _lib
    lib = _lib.load()
    assert lib.udaseg_gemm_1x1_bf16(3, 16, 64, 64, 4096, 4096, 4096, 0.0, None) != 0
    assert lib.udaseg_gemm_1x1_bf16(0, 16, 60, 64, 4096, 4096, 4096, 0.0, None) != 0
    assert lib.udaseg_gemm_1x1_bf16(0, 16, 64, 64, None, 4096, 4096, 0.0, None) != 0
    assert lib.udaseg_gemm_1x1_bf16(0, 16, 64, 64, 4096, 4096, 4096, 0.5, None) != 0
    assert not K.gemm_1x1_preferred(K.conv_desc(8, 192, 192, 64, 256, 1, 1, 0))       # the streaming kernel's layers
    assert not K.gemm_1x1_preferred(K.conv_desc(8, 48, 48, 256, 256, 3, 1, 1))


@pytest.mark.parametrize("n,h,w,c", [(8, 48, 48, 1024), (2, 7, 9, 64), (8, 96, 96, 512), (1, 5, 5, 2048), (3, 11, 13, 40)])
def test_bn_stats_bf16(K, n, h, w, c):
    """Per-channel sum / sum of squares of a bf16 tensor (the pass behind a library GEMM) against float64; the fp32 entry point
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
