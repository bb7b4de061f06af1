"""CPU oracle: domain discriminator, adversarial BCE loss, domain metrics, and the two train steps.

TEST INFRASTRUCTURE ONLY (see oracle/unet_ref.py header for who may import this).

Restates, with plain torch on CPU:
* ``DomainDiscriminator``        reference ``src/models/discriminator.py:4-56``
* ``AdversarialLoss``            reference ``src/models/losses.py:7-51``
* ``DomainAdaptationMetrics``    reference ``src/models/metrics.py:5-73`` (the parts the trainer calls)
* the source-only step           reference ``src/models/train.py:336-346``
* the adversarial iteration      reference ``src/models/adversarial_trainer.py:67-114``

Pinned: these reference modules import in the build container, so
``oracle/gen_golden.py`` runs the *reference's own* classes on seeded inputs and
commits the results to ``tests/golden/adversarial_ref.npz``; ``tests/test_oracle_adversarial.py``
requires this restatement to reproduce them bit for bit (CPU, same torch build).

Reference quirks reproduced on purpose:
* the discriminator ends in Sigmoid yet the loss is BCE-*with-logits*, i.e. the loss
  sees probabilities as logits;
* the discriminator consumes raw images, so the adversarial term adds nothing to the
  segmenter's gradients; its third forward still updates D's BN running statistics and
  its backward lands in D's ``.grad`` (zeroed at the start of the next iteration).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class DomainDiscriminatorRef(nn.Module):
    """4 x [conv4x4/2 p1 (+BN on 2..4) + LeakyReLU(0.2)], global-avg-pool, Linear(512,1), Sigmoid."""

    WIDTHS = (64, 128, 256, 512)

    def __init__(self, input_channels=3):
        super().__init__()
        layers = []
        cin = input_channels
        for i, cout in enumerate(self.WIDTHS):
            layers.append(nn.Conv2d(cin, cout, kernel_size=4, stride=2, padding=1))
            if i > 0:
                layers.append(nn.BatchNorm2d(cout))
            layers.append(nn.LeakyReLU(0.2, inplace=True))
            cin = cout
        # indices come out as features.{0,2,5,8} convs / {3,6,9} BNs, like the reference's state_dict
        self.features = nn.Sequential(*layers)
        self.classifier = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(cin, 1), nn.Sigmoid())

    def forward(self, x):
        return self.classifier(self.features(x))


def bce_with_logits_mean(x, y):
    """mean(softplus(x) - x*y) == nn.BCEWithLogitsLoss()(x, y)."""
    return F.binary_cross_entropy_with_logits(x, y)


class AdversarialLossRef:
    def __init__(self, lambda_adv=0.001):
        self.lambda_adv = lambda_adv

    def discriminator_loss(self, source_pred, target_pred):
        ls = bce_with_logits_mean(source_pred, torch.ones_like(source_pred))
        lt = bce_with_logits_mean(target_pred, torch.zeros_like(target_pred))
        return (ls + lt) / 2

    def generator_loss(self, target_pred):
        return self.lambda_adv * bce_with_logits_mean(target_pred, torch.ones_like(target_pred))


class DomainAdaptationMetricsRef:
    """Running domain accuracy + entropy of sigmoid(pred) (the reference re-applies sigmoid, F7)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.source_correct = self.source_total = 0
        self.target_correct = self.target_total = 0
        self.domain_entropy_sum = 0.0
        self.n_batches = 0

    def update(self, source_pred, target_pred):
        self.source_correct += int((source_pred >= 0.5).sum().item())
        self.source_total += source_pred.size(0)
        self.target_correct += int((target_pred < 0.5).sum().item())
        self.target_total += target_pred.size(0)
        p = torch.sigmoid(torch.cat([source_pred, target_pred], dim=0))
        ent = -p * torch.log(p + 1e-10) - (1 - p) * torch.log(1 - p + 1e-10)
        self.domain_entropy_sum += ent.mean().item()
        self.n_batches += 1

    def get_metrics(self):
        return {
            "source_domain_acc": f"{self.source_correct / max(self.source_total, 1):.4f}",
            "target_domain_acc": f"{self.target_correct / max(self.target_total, 1):.4f}",
            "domain_confusion": f"{self.domain_entropy_sum / max(self.n_batches, 1):.4f}",
        }


def segmentation_step(model, optimizer, images, masks, criterion=None):
    """One source-only iteration: zero_grad -> forward -> CE -> backward -> step (train.py:340-344)."""
    criterion = criterion or nn.CrossEntropyLoss()
    optimizer.zero_grad()
    outputs = model(images)
    loss = criterion(outputs, masks.long())
    loss.backward()
    optimizer.step()
    return loss.detach(), outputs.detach()


def adversarial_step(model, discriminator, adv_loss, optimizer, d_optimizer, source_images, source_masks,
                     target_images, metrics=None, criterion=None):
    """One adversarial iteration in the reference's order (adversarial_trainer.py:85-114)."""
    criterion = criterion or nn.CrossEntropyLoss()
    if source_masks.dim() == 4 and source_masks.size(1) == 1:
        source_masks = source_masks.squeeze(1)
    # --- discriminator update
    d_optimizer.zero_grad()
    p_s = discriminator(source_images)
    p_t = discriminator(target_images)
    if metrics is not None:
        metrics.update(p_s, p_t)
    d_loss = adv_loss.discriminator_loss(p_s, p_t)
    d_loss.backward()
    d_optimizer.step()
    # --- segmenter update
    optimizer.zero_grad()
    seg = model(source_images)
    seg_loss = criterion(seg, source_masks)
    p_t2 = discriminator(target_images)
    g_loss = adv_loss.generator_loss(p_t2)
    total = seg_loss + g_loss
    total.backward()
    optimizer.step()
    return {"seg_loss": seg_loss.detach(), "d_loss": d_loss.detach(), "adv_loss": g_loss.detach(),
            "total": total.detach(), "p_s": p_s.detach(), "p_t": p_t.detach()}


def synthetic_batch(n, h, w, classes=23, channels=3, seed=0, dtype=torch.float32):
    """The synthetic aerial batch every leg uses: images~N(0,1) seed, masks seed+1, target seed+2."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(n, channels, h, w, generator=g, dtype=dtype)
    g = torch.Generator().manual_seed(seed + 1)
    masks = torch.randint(0, classes, (n, h, w), generator=g, dtype=torch.int64)
    g = torch.Generator().manual_seed(seed + 2)
    target = torch.randn(n, channels, h, w, generator=g, dtype=dtype)
    return images, masks, target
