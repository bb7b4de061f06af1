"""``DomainDiscriminator`` -- mirror of reference ``src/models/discriminator.py:4-56`` on the HIP kernels.

Image-level domain classifier: 4 x [Conv 4x4 s2 p1 (+BatchNorm on layers 2-4) + LeakyReLU(0.2)] 3->64->128->256->512,
global average pool, Linear(512,1), Sigmoid -> probability ``[N,1]`` (0 = source, 1 = target).  ``state_dict`` keys
match the reference (``features.{0,2,5,8}.{weight,bias}``, ``features.{3,6,9}.*``, ``classifier.2.{weight,bias}``).
"""
import math

import torch
import torch.nn as nn

from . import kernels as K
from ._lib import ACT_LEAKY, require_gpu
from .engine import ArenaModule, BNP, ConvP, Plan, ceil4

SLOPE = 0.2


class LinearP(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(cin)
        nn.init.uniform_(self.bias, -bound, bound)


def _conv_default_init(conv):
    nn.init.kaiming_uniform_(conv.weight, a=math.sqrt(5))   # nn.Conv2d.reset_parameters
    bound = 1 / math.sqrt(conv.cin * conv.k * conv.k)
    nn.init.uniform_(conv.bias, -bound, bound)


class _DiscFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        p, tape = net._forward_plan(x, True)
        ctx.net, ctx.tape = net, tape
        return p

    @staticmethod
    def backward(ctx, dp):
        net, tape = ctx.net, ctx.tape
        ctx.tape = None
        net._backward_plan(tape, dp)              # delivers .grad itself (arena views)
        return (None, None) + (None,) * len(net._param_list)


class DomainDiscriminator(ArenaModule):
    def __init__(self, input_channels=3, compute_dtype=torch.float32):
        super().__init__()
        widths = (64, 128, 256, 512)
        layers, cin = [], input_channels
        for i, cout in enumerate(widths):
            conv = ConvP(cin, cout, 4, 2, 1, bias=True)
            _conv_default_init(conv)
            layers.append(conv)
            if i > 0:
                layers.append(BNP(cout))
            layers.append(nn.Identity())          # the LeakyReLU slot (fused into the kernels), keeps the indices
            cin = cout
        self.features = nn.Sequential(*layers)    # indices 0,2,5,8 convs / 3,6,9 BNs like the reference
        self.features[0].needs_dgrad = False      # raw images need no gradient
        self.classifier = nn.Sequential(nn.Identity(), nn.Identity(), LinearP(cin, 1), nn.Identity())
        self.input_channels = input_channels
        self.build_arena()
        if compute_dtype != torch.float32:
            self.set_compute_dtype(compute_dtype)

    def _layers(self):
        f = self.features
        return f[0], ((f[2], f[3]), (f[5], f[6]), (f[8], f[9])), self.classifier[2]

    def forward(self, x):
        require_gpu()
        if x.device.type != "cuda":
            raise RuntimeError("DomainDiscriminator.forward: input must live on the GPU (no CPU path in this build)")
        self.ensure_arena()
        x = x.float()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list):
            return _DiscFunction.apply(self, x, *self._param_list)
        with torch.no_grad():
            p, _ = self._forward_plan(x, False)
        return p

    def _forward_plan(self, x, save):
        P = Plan(self, self.training, save)
        conv0, blocks, lin = self._layers()
        x4 = K.nchw_to_nhwc(x, conv0.cin_p, P.st, dtype=P.adt)
        a0, d0 = P.conv(conv0, x4, ACT_LEAKY, SLOPE)     # bias + LeakyReLU fused into the conv epilogue
        h, recs = a0, []
        for conv, bn in blocks:
            h, rec = P.conv_bn_act(conv, bn, h, ACT_LEAKY, SLOPE)
            recs.append(rec)
        p, pooled = K.gap_linear_sigmoid_fwd(h, P.pvec(lin, "weight"), P.pvec(lin, "bias"), P.st)
        if self.training:
            self.tick_batchnorm_counters()
        if not save:
            return p, None
        return p, (P, (conv0, d0, x4, a0), recs, (lin, h, pooled, p))

    def _backward_plan(self, tape, dp):
        P, (conv0, d0, x4, a0), recs, (lin, h, pooled, p) = tape
        P.begin_backward()
        dz = torch.empty_like(h)
        K.gap_linear_sigmoid_bwd(dp.detach().float().contiguous(), p, pooled, P.pvec(lin, "weight"), dz,
                                 P.gvec(lin, "weight"), P.gvec(lin, "bias"), False, P.st)
        for rec in reversed(recs):
            x_in = rec[3]
            dx = torch.empty_like(x_in)
            P.conv_bn_act_bwd(rec, dz, dx=dx)
            dz = dx
        K.act_bwd(dz, a0, dz, ACT_LEAKY, SLOPE, P.st)    # through conv0's fused LeakyReLU, in place
        P.conv_bwd(conv0, d0, x4, dz, dx=None)
        P.join_side_stream()
        self.deliver_grads(P.garena)
