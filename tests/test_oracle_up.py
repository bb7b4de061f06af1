"""CPU: the phase decomposition of conv3x3(nearest_x2(a)) that csrc/conv_up_f32x3.hip computes (oracle/f32x3_ref.py) IS the
reference's op sequence -- F.interpolate(scale_factor=2, mode="nearest") then F.conv2d(padding=1), the ops of smp's DecoderBlock as
the trace fixture records them (aten::upsample_nearest2d, aten::cat, aten::_convolution) -- in exact arithmetic: float64 evaluations
of both agree to rounding, forward and gradient, for ragged sizes; and the fp32 pre-summed weights are within one rounding of the
exact sums."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.f32x3_ref import UP_TAPS, phase_weights, up_conv_phases, up_dgrad_phases


@pytest.mark.parametrize("n,h,w,ci,co", [(1, 1, 1, 2, 3), (2, 3, 5, 4, 6), (1, 8, 8, 16, 8), (1, 7, 4, 3, 5)])
def test_phase_form_equals_upsample_then_conv(n, h, w, ci, co):
    g = torch.Generator().manual_seed(h * 100 + w)
    a = torch.randn(n, ci, h, w, generator=g, dtype=torch.float64).requires_grad_(True)
    # weights exactly representable with room to spare: the fp32 pre-sums are then exact and the comparison is pure algebra
    wt = torch.randint(-64, 64, (co, ci, 3, 3), generator=g).double() / 32
    y = F.conv2d(F.interpolate(a, scale_factor=2, mode="nearest"), wt, padding=1)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    got = up_conv_phases(a.detach().numpy(), wt.float().numpy())
    assert np.abs(got - y.detach().numpy()).max() <= 1e-12 * max(1.0, float(y.detach().abs().max()))
    dgot = up_dgrad_phases(dy.numpy(), wt.float().numpy())
    assert np.abs(dgot - a.grad.numpy()).max() <= 1e-12 * max(1.0, float(a.grad.abs().max()))


def test_tap_sets_cover_every_kernel_row_once_per_phase():
    for ph in range(2):
        rows = sorted(UP_TAPS[(ph, 0)] + UP_TAPS[(ph, 1)])
        assert rows == [0, 1, 2]
    # 16 phase taps instead of 4 x 9 = 36 tap evaluations per 2 x 2 output pixels
    assert sum(1 for _ in UP_TAPS) * 4 == 16


def test_presummed_weights_are_one_rounding_from_exact():
    g = torch.Generator().manual_seed(3)
    wt = torch.randn(8, 16, 3, 3, generator=g)
    pw = phase_weights(wt.numpy())
    w64 = wt.double().numpy()
    for (py, u), ky in UP_TAPS.items():
        for (px, v), kx in UP_TAPS.items():
            exact = sum(w64[:, :, a, b] for a in ky for b in kx)
            mag = sum(np.abs(w64[:, :, a, b]) for a in ky for b in kx)
            # up to three fp32 additions: 3 half-units of the running sum, bounded by the magnitude sum
            assert (np.abs(pw[py, px, :, :, u, v].astype(np.float64) - exact) <= 3 * 2.0 ** -24 * mag + 1e-300).all()
