"""CPU oracle for the reference's segmentation loss family beyond plain cross entropy.

TEST INFRASTRUCTURE ONLY (see oracle/unet_ref.py header for who may import this).

Restates from first principles (explicit sums; no ``F.kl_div`` / ``F.cross_entropy`` / ``F.one_hot``), dtype-generic so
the tests can run it in float64:
* ``ConsistencyLossRef``            reference ``src/models/losses.py:53-108``
* ``DiceLossRef``                   reference ``src/models/losses.py:110-152``
* ``WeightedSegmentationLossRef``   reference ``src/models/losses.py:154-215``
* ``calculate_class_weights_ref``   reference ``src/models/losses.py:217-258``
* ``FineTuningLossRef``             reference ``src/models/losses.py:260-342``

Pinned: ``oracle/gen_golden.py`` runs the REFERENCE's own classes (importable in the build container) on seeded inputs,
requires this restatement to agree (float64: 1e-12 relative; gradients too) and commits inputs' seeds + outputs to
``tests/golden/losses_ref.npz``; ``tests/test_oracle_losses.py`` re-checks the restatement against that fixture.

Reference behaviours kept: 'batchmean' divides by the batch size only (not pixels); any reduction other than 'mean'
sums the focal term; ``FineTuningLoss`` applies ``domain_weight`` twice (as lambda_adv and as the multiplier).
"""
import torch

from .adversarial_ref import AdversarialLossRef


def _log_softmax(z):
    z = z - z.amax(dim=1, keepdim=True)
    return z - torch.log(torch.exp(z).sum(dim=1, keepdim=True))


def _one_hot(target, classes, dtype):
    b, h, w = target.shape
    oh = torch.zeros((b, classes, h, w), dtype=dtype, device=target.device)
    return oh.scatter_(1, target.long().unsqueeze(1), 1.0)


class ConsistencyLossRef:
    def __init__(self, temperature=0.5):
        self.temperature = temperature

    def __call__(self, pred1, pred2):
        l1 = _log_softmax(pred1 / self.temperature)
        l2 = _log_softmax(pred2 / self.temperature)
        p1, p2 = torch.exp(l1), torch.exp(l2)
        batch = pred1.shape[0]
        kl_2_given_1 = (p2 * (l2 - l1)).sum() / batch       # kl_div(input=log p1, target=p2, 'batchmean')
        kl_1_given_2 = (p1 * (l1 - l2)).sum() / batch
        return (kl_2_given_1 + kl_1_given_2) / 2


class DiceLossRef:
    def __init__(self, smooth=1.0):
        self.smooth = smooth

    def __call__(self, predictions, targets):
        p = torch.exp(_log_softmax(predictions))
        if targets.dim() == 3:
            targets = _one_hot(targets, predictions.shape[1], predictions.dtype)
        inter = (p * targets).sum(dim=(2, 3))
        union = p.sum(dim=(2, 3)) + targets.sum(dim=(2, 3))
        return 1.0 - ((2.0 * inter + self.smooth) / (union + self.smooth)).mean()


class WeightedSegmentationLossRef:
    def __init__(self, num_classes, class_weights=None, alpha=0.25, gamma=2.0, reduction='mean'):
        self.num_classes = num_classes
        self.class_weights = torch.ones(num_classes) if class_weights is None else class_weights
        self.alpha, self.gamma, self.reduction = alpha, gamma, reduction
        self.dice_loss = DiceLossRef()

    def focal_loss(self, inputs, targets):
        logp = _log_softmax(inputs)
        w = self.class_weights.to(inputs.dtype)[targets]
        ce = -w * logp.gather(1, targets.long().unsqueeze(1)).squeeze(1)
        pt = torch.exp(-ce)
        focal = self.alpha * (1 - pt) ** self.gamma * ce
        return focal.mean() if self.reduction == 'mean' else focal.sum()

    def __call__(self, inputs, targets, domain_weight=1.0):
        return domain_weight * (self.focal_loss(inputs, targets)
                                + self.dice_loss(inputs, _one_hot(targets, self.num_classes, inputs.dtype)))


def calculate_class_weights_ref(dataset, num_classes, method='effective_samples'):
    counts = torch.zeros(num_classes)
    for _, mask in dataset:
        for c in range(num_classes):
            counts[c] += (mask == c).sum().item()
    counts = torch.clamp(counts, min=1.0)
    if method == 'effective_samples':
        beta = 0.9999
        weights = (1.0 - beta) / (1.0 - torch.pow(beta, counts))
    else:
        weights = 1.0 / counts
    return weights / weights.sum() * num_classes


class FineTuningLossRef:
    def __init__(self, consistency_weight=1.0, domain_weight=0.1, supervised_weight=0.1, rampup_length=40, temperature=0.5):
        self.consistency_loss = ConsistencyLossRef(temperature)
        self.domain_loss = AdversarialLossRef(domain_weight)
        self.supervised_loss = DiceLossRef()
        self.consistency_weight, self.domain_weight = consistency_weight, domain_weight
        self.supervised_weight, self.rampup_length = supervised_weight, rampup_length

    def rampup(self, epoch):
        return 1.0 if epoch >= self.rampup_length else float(epoch) / self.rampup_length

    def __call__(self, pred1, pred2, domain_pred, epoch, supervised_pred=None, supervised_target=None):
        ramp = self.rampup(epoch)
        consistency = self.consistency_loss(pred1, pred2)
        domain_confusion = self.domain_loss.generator_loss(domain_pred)
        total = consistency * self.consistency_weight * ramp + domain_confusion * self.domain_weight * ramp
        supervised = torch.tensor(0.0)
        if supervised_pred is not None and supervised_target is not None:
            supervised = self.supervised_loss(supervised_pred, supervised_target.long())
            total = total + supervised * self.supervised_weight
        return {'total': total, 'consistency': consistency.detach(), 'domain_confusion': domain_confusion.detach(),
                'supervised': supervised.detach(), 'rampup_weight': torch.tensor(ramp)}


def loss_inputs(seed, batch=2, classes=23, h=12, w=10, dtype=torch.float64, scale=3.0):
    """Seeded inputs shared by the generator, the CPU tests and the GPU tests."""
    g = torch.Generator().manual_seed(seed)
    z1 = (torch.randn(batch, classes, h, w, generator=g, dtype=torch.float64) * scale).to(dtype)
    z2 = (z1.double() + torch.randn(batch, classes, h, w, generator=g, dtype=torch.float64)).to(dtype)
    target = torch.randint(0, classes, (batch, h, w), generator=g)
    target[0, 0, :3] = classes - 1                     # make sure the last class and a repeated class occur
    weights = (0.25 + 1.5 * torch.rand(classes, generator=g, dtype=torch.float64)).to(dtype)
    domain = torch.rand(batch, 1, generator=g, dtype=torch.float64).to(dtype)
    return z1, z2, target, weights, domain
