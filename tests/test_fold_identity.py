"""Pixel folding (csrc/conv_igemm.hip: fold_factor / fold_weights_bf16_kernel), the arithmetic restated on the CPU in float64:
a 3x3 / stride 1 / pad 1 convolution over [n][h][w][ci] equals the same kind of convolution over units of F pixels,
[n][h][w/F][F*ci] -> [n][h][w/F][F*co], with  W'[(fo,o)][r][qu][(fi,c)] = W[o][r][F(qu-1)+fi-fo+1][c]  (zero outside 0..2);
and the data gradient is that construction on the flipped, transposed weights.  No GPU: this pins the FORMULA the kernel
implements; tests/test_gpu_bf16.py pins the kernel."""
import pytest
import torch
import torch.nn.functional as F_


def fold_weights(w_ptc, F, flip):
    """w_ptc: [cp][3][3][cg] (produced channel, tap row, tap column, gathered channel) -> [F*cp][3][3][F*cg]."""
    cp, _, _, cg = w_ptc.shape
    out = torch.zeros(F, cp, 3, 3, F, cg, dtype=w_ptc.dtype)
    for fo in range(F):
        for fi in range(F):
            for qu in range(3):
                dx = F * (qu - 1) + fi - fo + 1
                if 0 <= dx <= 2:
                    for r in range(3):
                        out[fo, :, r, qu, fi, :] = w_ptc[:, 2 - r, 2 - dx, :] if flip else w_ptc[:, r, dx, :]
    return out.reshape(F * cp, 3, 3, F * cg)


def conv_nhwc(x, w_ptc):
    """x [n][h][w][cg], w [cp][3][3][cg] -> [n][h][w][cp], 3x3 / stride 1 / pad 1."""
    y = F_.conv2d(x.permute(0, 3, 1, 2), w_ptc.permute(0, 3, 1, 2), padding=1)
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("F,ci,co,h,w", [(2, 32, 16, 5, 8), (2, 32, 32, 4, 6), (4, 16, 16, 3, 8), (4, 16, 24, 6, 12)])
def test_folded_forward_equals_direct(F, ci, co, h, w):
    g = torch.Generator().manual_seed(F * 100 + ci + co)
    x = torch.randn(2, h, w, ci, generator=g, dtype=torch.float64)
    wt = torch.randn(co, 3, 3, ci, generator=g, dtype=torch.float64)
    y = conv_nhwc(x, wt)
    yf = conv_nhwc(x.reshape(2, h, w // F, F * ci), fold_weights(wt, F, False))      # zero-copy reinterpretation of x
    assert torch.allclose(yf.reshape(2, h, w, co), y, rtol=0, atol=1e-12)


@pytest.mark.parametrize("F,ci,co,h,w", [(2, 16, 32, 5, 8), (4, 16, 16, 4, 8), (4, 32, 16, 3, 12)])
def test_folded_data_gradient_equals_autograd(F, ci, co, h, w):
    """dx of a conv ci -> co: gathers dy (co channels; the fold is 64 / co), produces ci channels, from the dgrad packing
    w_t[ci][tap][co] read as the flipped forward convolution."""
    g = torch.Generator().manual_seed(F * 10 + ci + co)
    x = torch.randn(2, h, w, ci, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(co, 3, 3, ci, generator=g, dtype=torch.float64)
    dy = torch.randn(2, h, w, co, generator=g, dtype=torch.float64)
    conv_nhwc(x, wt).backward(dy)
    w_t = wt.permute(3, 1, 2, 0).contiguous()                      # [ci][3][3][co]: what udaseg_pack_dgrad_* produce
    dxf = conv_nhwc(dy.reshape(2, h, w // F, F * co), fold_weights(w_t, F, True))
    assert torch.allclose(dxf.reshape(2, h, w, ci), x.grad, rtol=0, atol=1e-12)
