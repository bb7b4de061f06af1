"""CPU oracle for the feature-level domain-adaptation design.

TEST INFRASTRUCTURE ONLY (see oracle/unet_ref.py header for who may import this).

Restates with plain torch on CPU, dtype-generic (tests run it in float64):
* ``FeatureDiscriminatorRef``   reference ``src/models/uda.py:8-28`` (class ``DomainDiscriminator`` there)
* ``UDASegmentationModelRef``   reference ``src/models/uda.py:30-76`` over ``oracle.unet_ref.UnetRef``
* ``smp_multiclass_dice``       ``segmentation_models_pytorch.losses.DiceLoss(mode='multiclass')`` (third-party; the
                                reference constructs it at ``src/models/uda.py:84``)
* ``UDALossRef``                reference ``src/models/uda.py:80-97``
* ``gradient_reverse_ref``      reference ``src/models/uda.py:99-111``
* ``phase2_step_ref``           reference ``src/models/trainer_phases.py:136-164``

PARITY UNPINNED: ``src/models/uda.py`` imports ``segmentation_models_pytorch`` at module level, which is not installed
here (an ordinary ModuleNotFoundError), so neither the reference module nor smp's DiceLoss can be run to
produce vectors, and the reference's tests never exercise this file (dead alternate design).  The Dice term
follows smp's published algorithm (requirements.txt:3 asks ``segmentation-models-pytorch>=0.3.0``, unpinned):
softmax probabilities; per class, over batch and pixels, ``score = (2*I + smooth) / clamp_min(sum(p) + sum(onehot) + smooth,
eps)`` with ``smooth=0, eps=1e-7``; ``loss = mean_c((1 - score_c) * [class c present in the targets])``.
Everything else here is ``torch.nn`` layers composed as the reference's source reads.
"""
import torch
import torch.nn as nn

from .unet_ref import UnetRef

ENCODER_TOP = {"resnet18": 512, "resnet34": 512, "resnet50": 2048}


class FeatureDiscriminatorRef(nn.Module):
    def __init__(self, num_channels=512):
        super().__init__()
        mods, cin = [], num_channels
        for cout in (512, 256, 128):
            mods += [nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout), nn.ReLU()]
            cin = cout
        mods += [nn.Conv2d(cin, 1, kernel_size=1), nn.AdaptiveAvgPool2d(1)]
        self.discriminator = nn.Sequential(*mods)

    def forward(self, x):
        return self.discriminator(x)


class _GradReverse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -ctx.alpha * g, None


def gradient_reverse_ref(x, alpha):
    return _GradReverse.apply(x, alpha)


class UDASegmentationModelRef(nn.Module):
    def __init__(self, encoder_name="resnet50", classes=23, head_in_da_forward=False, grl_alpha=None):
        super().__init__()
        self.segmentation_model = UnetRef(encoder_name, classes=classes)
        self.domain_discriminator = FeatureDiscriminatorRef(ENCODER_TOP[encoder_name])
        self.head_in_da_forward, self.grl_alpha = head_in_da_forward, grl_alpha

    def forward(self, x, domain_adaptation=False):
        m = self.segmentation_model
        if not domain_adaptation:
            return m(x)
        features = m.encoder(x)
        seg = m.decoder(*features)                    # upstream stops here: no segmentation head
        if self.head_in_da_forward:
            seg = m.segmentation_head(seg)
        top = features[-1]
        if self.grl_alpha is not None:
            top = gradient_reverse_ref(top, self.grl_alpha)
        return seg, self.domain_discriminator(top).squeeze(-1).squeeze(-1)


def smp_multiclass_dice(y_pred, y_true, smooth=0.0, eps=1e-7):
    bs, c = y_pred.shape[:2]
    p = torch.log_softmax(y_pred, dim=1).exp().reshape(bs, c, -1)
    t = y_true.reshape(bs, -1)
    onehot = torch.zeros_like(p).scatter_(1, t.unsqueeze(1), 1.0)
    inter = (p * onehot).sum(dim=(0, 2))
    card = (p + onehot).sum(dim=(0, 2))
    score = (2.0 * inter + smooth) / (card + smooth).clamp_min(eps)
    present = (onehot.sum(dim=(0, 2)) > 0).to(p.dtype)
    return ((1.0 - score) * present).mean()


def bce_with_logits(x, y):
    return (torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))).mean()


class UDALossRef:
    def __init__(self, lambda_adv=0.001):
        self.lambda_adv = lambda_adv
        self.domain_loss = bce_with_logits
        self.segmentation_loss = smp_multiclass_dice

    def __call__(self, pred, target, domain_pred=None, domain_target=None):
        seg = self.segmentation_loss(pred, target)
        if domain_pred is not None and domain_target is not None:
            return seg + self.lambda_adv * self.domain_loss(domain_pred, domain_target)
        return seg


def phase2_step_ref(model, criterion, optimizer, source_images, source_masks, target_images):
    optimizer.zero_grad()
    source_seg, source_domain = model(source_images, domain_adaptation=True)
    _, target_domain = model(target_images, domain_adaptation=True)
    batch = source_images.size(0)
    ones = torch.ones(batch, dtype=source_domain.dtype)
    zeros = torch.zeros(batch, dtype=source_domain.dtype)
    seg_loss = criterion(source_seg, source_masks.long())
    domain_loss = (criterion.domain_loss(source_domain.view(batch), ones)
                   + criterion.domain_loss(target_domain.view(batch), zeros)) / 2
    total = seg_loss + criterion.lambda_adv * domain_loss
    total.backward()
    optimizer.step()
    return total, seg_loss, domain_loss
