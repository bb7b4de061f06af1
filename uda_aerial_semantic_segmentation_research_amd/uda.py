"""Feature-level domain adaptation pieces -- the build's counterpart of reference ``src/models/uda.py:8-111`` and of the
phase-2 iteration of ``src/models/trainer_phases.py:104-208``.

Unlike the image-level ``discriminator.DomainDiscriminator`` (whose loss never reaches the segmenter), this
discriminator reads the deepest encoder feature, so the domain loss trains the encoder: its gradient enters the segmenter's
backward plan at ``encoder(x)[-1]`` (``Unet.forward_parts(..., "features")``).

* ``DomainDiscriminator(num_channels=512)``  3 x [Conv3x3 p1 (bias) + BatchNorm + ReLU] C->512->256->128, Conv1x1 128->1,
  global average pool -> ``[N,1,1,1]`` LOGITS (``state_dict`` keys ``discriminator.{0,1,3,4,6,7,9}.*`` as upstream).
* ``UDASegmentationModel``  wraps ``Unet`` + that discriminator; ``forward(x, domain_adaptation=True)`` returns
  ``(decoder output, domain logits [N,1])``.  Upstream returns the decoder's 16-channel output there, not the logits
  (SURVEY F10); that is kept as the default and ``head_in_da_forward=True`` selects the segmentation logits instead.
  ``grl_alpha`` (default None = upstream: no reversal) routes the feature through ``gradient_reverse_layer`` first.
* ``UDALoss(lambda_adv)``  smp multiclass Dice + lambda_adv x BCE-with-logits, ``gradient_reverse_layer`` /
  ``GradientReverseFunction`` (defined upstream, called nowhere), and ``phase2_step`` -- one phase-2 iteration.
"""
import torch
import torch.nn as nn

from . import kernels as K
from ._lib import ACT_LEAKY, require_gpu
from .discriminator import _conv_default_init
from .engine import ArenaModule, BNP, ConvP, Plan
from .losses import BCEWithLogitsLoss, MulticlassDiceLoss
from .unet import ENCODERS, Unet

RELU = (ACT_LEAKY, 0.0)


class _FeatureDiscFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, feat, *params):
        out, tape = net._forward_plan(feat, True)
        ctx.net, ctx.tape = net, tape
        return out

    @staticmethod
    def backward(ctx, d_out):
        net, tape = ctx.net, ctx.tape
        ctx.tape = None
        d_feat = net._backward_plan(tape, d_out, ctx.needs_input_grad[1])
        return (None, d_feat) + (None,) * len(net._param_list)


class DomainDiscriminator(ArenaModule):
    """Domain discriminator on encoder features (fp32)."""

    def __init__(self, num_channels=512):
        super().__init__()
        layers, cin = [], num_channels
        for cout in (512, 256, 128):
            conv = ConvP(cin, cout, 3, 1, 1, bias=True)
            _conv_default_init(conv)
            layers += [conv, BNP(cout), nn.Identity()]          # the ReLU slot keeps upstream's Sequential indices
            cin = cout
        last = ConvP(cin, 1, 1, 1, 0, bias=True)
        _conv_default_init(last)
        layers += [last, nn.Identity()]                         # 9: Conv1x1, 10: the pooling slot
        self.discriminator = nn.Sequential(*layers)
        self.num_channels = num_channels
        self.build_arena()

    def _layers(self):
        d = self.discriminator
        return ((d[0], d[1]), (d[3], d[4]), (d[6], d[7])), d[9]

    def forward(self, x):
        require_gpu()
        if x.device.type != "cuda":
            raise RuntimeError("uda.DomainDiscriminator.forward: input must live on the GPU (no CPU path in this build)")
        if x.dim() != 4 or x.shape[1] != self.num_channels:
            raise ValueError(f"expected features [N,{self.num_channels},h,w], got {tuple(x.shape)}")
        self.ensure_arena()
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self._param_list)):
            return _FeatureDiscFunction.apply(self, x, *self._param_list)
        with torch.no_grad():
            out, _ = self._forward_plan(x, False)
        return out

    @staticmethod
    def _as_nhwc(x):
        """[N,C,h,w] -> dense NHWC fp32 (zero-copy for the strided views ``Unet.forward_parts`` hands out)."""
        n, c, h, w = x.shape
        if (c % 4 == 0 and x.dtype == torch.float32 and x.stride() == (h * w * c, 1, w * c, c)):
            return x.detach().permute(0, 2, 3, 1)
        return K.nchw_to_nhwc(x.detach().float().contiguous(), (c + 3) // 4 * 4)

    def _forward_plan(self, x, save):
        P = Plan(self, self.training, save)
        blocks, last = self._layers()
        h, recs = self._as_nhwc(x), []
        for conv, bn in blocks:
            h, rec = P.conv_bn_act(conv, bn, h, *RELU)
            recs.append(rec)
        # Conv1x1(128 -> 1) then the spatial mean == mean then a 128 -> 1 dot product: row 0 of the conv's physical weight
        w_row = P.w(last).view(last.cout_p, -1)[0]
        logit, pooled = K.gap_linear_fwd(h, w_row, P.b(last), P.st)
        if self.training:
            self.tick_batchnorm_counters()
        out = logit.view(-1, 1, 1, 1)
        if not save:
            return out, None
        return out, (P, recs, (last, h, pooled))

    def _backward_plan(self, tape, d_out, need_input_grad):
        P, recs, (last, h, pooled) = tape
        P.begin_backward()
        dz = torch.empty_like(h)
        w_row = P.w(last).view(last.cout_p, -1)[0]
        gw_row = P.gw(last).view(last.cout_p, -1)[0]
        K.gap_linear_bwd(d_out.detach().float().contiguous().view(-1), pooled, w_row, dz, gw_row, P.gvec(last, "bias"), False, P.st)
        d_feat = None
        for i, rec in enumerate(reversed(recs)):
            x_in = rec[3]
            first = i == len(recs) - 1
            dx = torch.empty_like(x_in) if (need_input_grad or not first) else None
            P.conv_bn_act_bwd(rec, dz, dx=dx)
            dz = dx
        if need_input_grad:
            d_feat = dz.permute(0, 3, 1, 2)[:, : self.num_channels]
        P.join_side_stream()
        self.deliver_grads(P.garena)
        return d_feat


class GradientReverseFunction(torch.autograd.Function):
    """Identity in the forward pass; multiplies the gradient by ``-alpha`` on the way back."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        if not grad_output.is_cuda:
            raise RuntimeError("gradient_reverse_layer: gradients must live on the GPU (no CPU path in this build)")
        g = grad_output.float()
        g = g if g.is_contiguous() or _dense_permuted(g) else g.contiguous()
        out = torch.empty_like(g)                               # preserves g's (dense) strides
        K.scale(g, -float(ctx.alpha), out)
        return out.to(grad_output.dtype), None


def _dense_permuted(t):
    return t.is_contiguous(memory_format=torch.channels_last) if t.dim() == 4 else False


def gradient_reverse_layer(x, alpha):
    return GradientReverseFunction.apply(x, alpha)


class UDASegmentationModel(nn.Module):
    def __init__(self, encoder_name="resnet50", encoder_weights="imagenet", classes=23, activation=None,
                 head_in_da_forward=False, grl_alpha=None):
        super().__init__()
        self.segmentation_model = Unet(encoder_name=encoder_name, encoder_weights=encoder_weights, in_channels=3,
                                       classes=classes, activation=activation)
        self.domain_discriminator = DomainDiscriminator(num_channels=ENCODERS[encoder_name][2][-1])
        self.encoder_name = encoder_name
        self.head_in_da_forward = head_in_da_forward
        self.grl_alpha = grl_alpha

    def forward(self, x, domain_adaptation=False):
        if not domain_adaptation:
            return self.segmentation_model(x)
        seg_name = "logits" if self.head_in_da_forward else "decoder"
        seg, feat = self.segmentation_model.forward_parts(x, (seg_name, "features"))
        if self.grl_alpha is not None:
            feat = gradient_reverse_layer(feat, self.grl_alpha)
        domain = self.domain_discriminator(feat)
        return seg, domain.squeeze(-1).squeeze(-1)

    def get_encoder_features(self, x):
        """The deepest encoder feature ``[N, C, H/32, W/32]``."""
        return self.segmentation_model.forward_parts(x, ("features",))


class UDALoss(nn.Module):
    """Dice (smp multiclass form) on the segmentation output, plus ``lambda_adv`` x BCE-with-logits on the domain logits
    when both ``domain_pred`` and ``domain_target`` are given."""

    def __init__(self, lambda_adv=0.001):
        super().__init__()
        self.segmentation_loss = MulticlassDiceLoss()
        self.domain_loss = BCEWithLogitsLoss()
        self.lambda_adv = lambda_adv

    def forward(self, pred, target, domain_pred=None, domain_target=None):
        seg_loss = self.segmentation_loss(pred, target)
        if domain_pred is not None and domain_target is not None:
            return seg_loss + self.lambda_adv * self.domain_loss(domain_pred, domain_target)
        return seg_loss


def phase2_step(model, criterion, optimizer, source_images, source_masks, target_images):
    """One supervised-adversarial iteration in upstream's order (``src/models/trainer_phases.py:136-164``): both domains
    through the model with ``domain_adaptation=True``, Dice on the source output, the mean of the two domain BCE terms
    (source label 1, target label 0) weighted by ``criterion.lambda_adv``, one backward, one optimizer step.
    Returns (total, seg_loss, domain_loss) as device tensors."""
    optimizer.zero_grad()
    source_seg, source_domain = model(source_images, domain_adaptation=True)
    _, target_domain = model(target_images, domain_adaptation=True)
    batch = source_images.size(0)
    ones = torch.ones(batch, device=source_images.device)
    zeros = torch.zeros(batch, device=source_images.device)
    seg_loss = criterion(source_seg, source_masks.long())
    domain_loss = (criterion.domain_loss(source_domain.view(batch), ones)
                   + criterion.domain_loss(target_domain.view(batch), zeros)) / 2
    total = seg_loss + criterion.lambda_adv * domain_loss
    total.backward()
    optimizer.step()
    return total, seg_loss, domain_loss
