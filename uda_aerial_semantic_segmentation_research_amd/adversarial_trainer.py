"""``AdversarialTrainer`` -- the build's counterpart of reference ``src/models/adversarial_trainer.py:9-218``: same
constructor and ``train_epoch`` / ``validate`` / ``train`` signatures and return shapes, same iteration order
(``:85-114``): D step on (source, target) -> segmenter step with CE on source + lambda * BCE(D(target), 1).

Reference behaviours kept as they are: the discriminator sees raw images, so the adversarial
term adds nothing to the segmenter's gradients but its third forward still updates D's BatchNorm statistics and its
backward lands in D's ``.grad``; D's optimizer is created lazily with the segmenter optimizer's lr (``:55-59``); the
target loader is cycled (``:69-73``); ``[B,1,H,W]`` masks are squeezed (``:80-82``).
"""
import torch

from .discriminator import DomainDiscriminator
from .losses import AdversarialLoss
from .metrics import DomainAdaptationMetrics
from .optim import FusedAdam
from .train import SegmentationTrainer


class AdversarialTrainer(SegmentationTrainer):
    def __init__(self, model, device, lambda_adv=0.001):
        """model: segmentation model; device; lambda_adv: weight of the adversarial loss."""
        super().__init__(model, device)
        self.discriminator = DomainDiscriminator().to(device)
        self.adversarial_loss = AdversarialLoss(lambda_adv)
        self.discriminator_optimizer = None
        self.domain_metrics = DomainAdaptationMetrics()
        self.d_grad_reducer = None
        self.last_losses = {}

    def calculate_iou(self, pred, target):
        """Binary-style IoU of the argmax mask against the target mask (reference :25-39)."""
        inter = torch.logical_and(pred, target)
        union = torch.logical_or(pred, target)
        return (torch.sum(inter).float() / (torch.sum(union).float() + 1e-8)).item()

    def adversarial_step(self, source_images, source_masks, target_images, optimizer, update_metrics=True):
        """One iteration of reference :85-114; returns loss tensors (no host sync besides the optional metrics)."""
        if source_masks.dim() == 4 and source_masks.size(1) == 1:
            source_masks = source_masks.squeeze(1)
        # ---- train discriminator
        self.discriminator_optimizer.zero_grad()
        source_domain_pred = self.discriminator(source_images)
        target_domain_pred = self.discriminator(target_images)
        if update_metrics:
            self.domain_metrics.update(source_domain_pred, target_domain_pred)
        d_loss = self.adversarial_loss.discriminator_loss(source_domain_pred, target_domain_pred)
        d_loss.backward()
        if self.d_grad_reducer is not None:
            self.d_grad_reducer.allreduce_now(self.discriminator)
        self.discriminator_optimizer.step()
        # ---- train segmentation model
        optimizer.zero_grad()
        source_seg_pred = self.model(source_images)
        seg_loss = self.criterion(source_seg_pred, source_masks)
        target_domain_pred = self.discriminator(target_images)
        adv_loss = self.adversarial_loss.generator_loss(target_domain_pred)
        total_g_loss = seg_loss + adv_loss
        total_g_loss.backward()
        if self.grad_reducer is not None:
            self.grad_reducer.finish()
        optimizer.step()
        return seg_loss, d_loss, adv_loss, total_g_loss

    def train_epoch(self, source_dataloader, target_dataloader, optimizer, epoch):
        """Train one epoch on source (labelled) + target (unlabelled) data; returns (mean loss, domain metrics)."""
        self.model.train()
        self.discriminator.train()
        self.domain_metrics.reset()
        if self.discriminator_optimizer is None:
            self.discriminator_optimizer = FusedAdam(self.discriminator.parameters(), lr=optimizer.param_groups[0]["lr"])
        total_loss = 0.0
        target_iter = iter(target_dataloader)
        for batch_idx, (source_images, source_masks) in enumerate(source_dataloader):
            try:
                target_images = next(target_iter)
            except StopIteration:
                target_iter = iter(target_dataloader)
                target_images = next(target_iter)
            source_images = source_images.to(self.device)
            source_masks = source_masks.to(self.device)
            target_images = target_images.to(self.device)
            seg_loss, d_loss, adv_loss, total = self.adversarial_step(source_images, source_masks, target_images, optimizer)
            host = torch.stack([seg_loss.detach(), d_loss.detach(), adv_loss.detach(), total.detach()]).cpu().tolist()
            self.last_losses = {"seg_loss": host[0], "d_loss": host[1], "adv_loss": host[2], "total": host[3]}
            total_loss += host[3]
        return total_loss / len(source_dataloader), self.domain_metrics.get_metrics()

    def validate(self, dataloader):
        """Validate; returns (mean loss, {'iou','accuracy'} formatted like the reference)."""
        self.model.eval()
        total_loss = total_iou = total_accuracy = 0.0
        with torch.no_grad():
            for images, masks in dataloader:
                images = images.to(self.device)
                masks = masks.to(self.device)
                if masks.dim() == 4 and masks.size(1) == 1:
                    masks = masks.squeeze(1)
                outputs = self.model(images)
                loss = self.criterion(outputs, masks)
                pred_masks = outputs.argmax(dim=1)
                total_iou += self.calculate_iou(pred_masks, masks)
                total_accuracy += (pred_masks == masks).float().mean().item()
                total_loss += loss.item()
        n = len(dataloader)
        return total_loss / n, {"iou": f"{total_iou / n:.4f}", "accuracy": f"{total_accuracy / n:.4f}"}

    def train(self, source_dataloader, target_dataloader, valid_dataloader, epochs, learning_rate, patience=3):
        """Train with domain adaptation; plain patience early stopping on the validation loss."""
        optimizer = FusedAdam(self.model.parameters(), lr=learning_rate)
        best_valid_loss = float("inf")
        patience_counter = 0
        for epoch in range(1, epochs + 1):
            train_loss, domain_metrics = self.train_epoch(source_dataloader, target_dataloader, optimizer, epoch)
            valid_loss, valid_metrics = self.validate(valid_dataloader)
            print(f"Train Loss: {train_loss:.4f}")
            print(f"Valid Loss: {valid_loss:.4f}")
            print(f"Valid Metrics: {valid_metrics}")
            print(f"Domain Metrics: {domain_metrics}")
            if valid_loss < best_valid_loss:
                best_valid_loss = valid_loss
                patience_counter = 0
            else:
                patience_counter += 1
                if patience_counter >= patience:
                    print(f"Early stopping after {epoch} epochs")
                    break
