"""``AdversarialTrainer`` -- the build's counterpart of reference ``src/models/adversarial_trainer.py:9-218``.

Same public surface (constructor ``(model, device, lambda_adv=0.001)``; ``train_epoch(source_dataloader,
target_dataloader, optimizer, epoch) -> (mean loss, domain metrics)``; ``validate(dataloader) -> (loss, {'iou',
'accuracy'})``; ``train(...)``; attributes ``discriminator``, ``adversarial_loss``, ``domain_metrics``,
``discriminator_optimizer``) and the same iteration order (reference ``:85-114``): a discriminator update on one source
and one target batch, then a segmenter update with cross entropy on the source batch plus ``lambda_adv`` x BCE of the
discriminator's verdict on the target batch.

Reference behaviours kept as they are: the discriminator judges raw images, so the adversarial
term contributes nothing to the segmenter's gradients, yet its third forward still moves D's BatchNorm statistics and its
backward lands in D's ``.grad``; D's optimizer appears lazily with the segmenter optimizer's learning rate (``:55-59``);
a shorter target loader wraps around (``:69-73``); ``[B,1,H,W]`` masks lose their channel axis (``:80-82``).
"""
import torch

from .discriminator import DomainDiscriminator
from .losses import AdversarialLoss
from .metrics import DomainAdaptationMetrics
from .optim import FusedAdam
from .train import SegmentationTrainer


def _wrap_around(loader):
    """Yield batches forever, restarting the loader whenever it runs dry."""
    while True:
        empty = True
        for batch in loader:
            empty = False
            yield batch
        if empty:
            raise ValueError("target dataloader yields no batches")


def _drop_channel_axis(masks):
    return masks.squeeze(1) if masks.dim() == 4 and masks.size(1) == 1 else masks


class AdversarialTrainer(SegmentationTrainer):
    def __init__(self, model, device, lambda_adv=0.001):
        super().__init__(model, device)
        self.discriminator = DomainDiscriminator(compute_dtype=getattr(model, "compute_dtype", torch.float32)).to(device)
        self.adversarial_loss = AdversarialLoss(lambda_adv)
        self.domain_metrics = DomainAdaptationMetrics()
        self.discriminator_optimizer = None      # created on first use, see train_epoch
        self.d_grad_reducer = None               # ddp.GradAllReducer (class) when data-parallel
        self.last_losses = {}

    # ------------------------------------------------------------------------------------------------ one iteration
    def adversarial_step(self, source_images, source_masks, target_images, optimizer, update_metrics=True):
        """The timed hot path of this trainer.  Returns (seg_loss, d_loss, adv_loss, total) as device tensors."""
        D, L = self.discriminator, self.adversarial_loss
        labels = _drop_channel_axis(source_masks)

        # (1) discriminator: source should score 1, target 0
        self.discriminator_optimizer.zero_grad()
        verdict_src, verdict_tgt = D(source_images), D(target_images)
        if update_metrics:
            self.domain_metrics.update(verdict_src, verdict_tgt)
        d_loss = L.discriminator_loss(verdict_src, verdict_tgt)
        d_loss.backward()
        if self.d_grad_reducer is not None:
            self.d_grad_reducer.allreduce_now(D)
        self.discriminator_optimizer.step()

        # (2) segmenter: supervised on source, "fool D" term on target
        optimizer.zero_grad()
        seg_loss = self.criterion(self.model(source_images), labels)
        adv_loss = L.generator_loss(D(target_images))
        total = seg_loss + adv_loss
        total.backward()
        if self.grad_reducer is not None:
            self.grad_reducer.finish()
        optimizer.step()
        return seg_loss, d_loss, adv_loss, total

    # ------------------------------------------------------------------------------------------------------ epochs
    def train_epoch(self, source_dataloader, target_dataloader, optimizer, epoch):
        for net in (self.model, self.discriminator):
            net.train()
        self.domain_metrics.reset()
        if self.discriminator_optimizer is None:
            self.discriminator_optimizer = FusedAdam(self.discriminator.parameters(), lr=optimizer.param_groups[0]["lr"])
        running, batches = 0.0, 0
        for (images, masks), target in zip(source_dataloader, _wrap_around(target_dataloader)):
            losses = self.adversarial_step(images.to(self.device), masks.to(self.device), target.to(self.device), optimizer)
            seg, dl, adv, tot = torch.stack([t.detach() for t in losses]).tolist()     # one transfer per iteration
            self.last_losses = {"seg_loss": seg, "d_loss": dl, "adv_loss": adv, "total": tot}
            running += tot
            batches += 1
        return running / max(batches, 1), self.domain_metrics.get_metrics()

    def calculate_iou(self, pred, target):
        """Binary-style IoU of two index masks: |pred AND target| / |pred OR target| (reference ``:25-39``)."""
        hit = torch.logical_and(pred, target).sum().float()
        any_ = torch.logical_or(pred, target).sum().float()
        return (hit / (any_ + 1e-8)).item()

    def validate(self, dataloader):
        self.model.eval()
        sums = torch.zeros(3, dtype=torch.float64, device=self.device)      # loss, iou, accuracy
        n = 0
        with torch.no_grad():
            for images, masks in dataloader:
                images = images.to(self.device)
                labels = _drop_channel_axis(masks.to(self.device))
                logits = self.model(images)
                pred = logits.argmax(dim=1)
                hit = torch.logical_and(pred, labels).sum().double()
                any_ = torch.logical_or(pred, labels).sum().double()
                sums += torch.stack([self.criterion(logits, labels).double(), hit / (any_ + 1e-8),
                                     (pred == labels).double().mean()])
                n += 1
        loss, iou, acc = (sums / max(n, 1)).tolist()
        return loss, {"iou": f"{iou:.4f}", "accuracy": f"{acc:.4f}"}

    def train(self, source_dataloader, target_dataloader, valid_dataloader, epochs, learning_rate, patience=3):
        optimizer = FusedAdam(self.model.parameters(), lr=learning_rate)
        best, stale = float("inf"), 0
        for epoch in range(1, epochs + 1):
            train_loss, dm = self.train_epoch(source_dataloader, target_dataloader, optimizer, epoch)
            valid_loss, vm = self.validate(valid_dataloader)
            for label, value in (("Train Loss", f"{train_loss:.4f}"), ("Valid Loss", f"{valid_loss:.4f}"),
                                 ("Valid Metrics", vm), ("Domain Metrics", dm)):
                print(f"{label}: {value}")
            stale = 0 if valid_loss < best else stale + 1
            best = min(best, valid_loss)
            if stale >= patience:                # plain patience on the validation loss (reference :211-218)
                print(f"Early stopping after {epoch} epochs")
                break
