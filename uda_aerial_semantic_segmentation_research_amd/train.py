"""``SegmentationTrainer`` -- the build's counterpart of reference ``src/models/train.py::SegmentationTrainer``
(``:197-503``): same constructor, ``train_epoch`` / ``validate`` / ``train`` / ``calculate_metrics`` signatures and
return shapes, same step order (``:336-346``: zero_grad -> forward -> CrossEntropy -> backward -> Adam step), with the
model, loss and optimizer running on the HIP kernels.  ``EarlyStopping`` restates ``:79-195``.

Left out on purpose (OUT OF SCOPE): the sklearn confusion-matrix / ROC / PR figure logging
(``:245-328,365-387``) and the torchmetrics objects; TensorBoard logging degrades to a no-op when the package is absent.
"""
import os
from pathlib import Path

import numpy as np
import torch

from .config import Config
from .losses import CrossEntropyLoss
from .metrics import segmentation_metrics
from .optim import FusedAdam


class NullLogger:
    """Stand-in for the reference's TensorboardLogger (src/visualization/tensorboard_logger.py:11-86)."""

    def __init__(self, log_dir=None):
        self.log_dir = log_dir
        self.scalars = {}

    def log_scalar(self, tag, value, step):
        self.scalars.setdefault(tag, []).append((step, float(value)))

    def log_scalars(self, main_tag, tag_scalar_dict, step):
        for k, v in tag_scalar_dict.items():
            self.log_scalar(f"{main_tag}/{k}", v, step)

    def log_image(self, *a, **k):
        pass

    log_images = log_figure = log_histogram = log_model_graph = log_image

    def close(self):
        pass


class EarlyStopping:
    """Weighted multi-metric early stopping (reference train.py:79-195), including its ``min_epochs`` gate."""

    def __init__(self, patience=7, min_delta=0.0, mode="min", min_epochs=10, metrics_to_track=None, weights=None,
                 verbose=False):
        self.patience, self.min_delta, self.mode, self.min_epochs = patience, min_delta, mode, min_epochs
        self.metrics_to_track = metrics_to_track or ["loss"]
        self.weights = weights or {"loss": 1.0}
        self.verbose = verbose
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.best_metrics = {}
        self.val_loss_min = float("inf")
        self.metric_history = {m: [] for m in self.metrics_to_track}

    def _calculate_score(self, metrics):
        return sum(self.weights[m] * v for m, v in metrics.items() if m in self.weights)

    def _is_better(self, current, best):
        return current < best - self.min_delta if self.mode == "min" else current > best + self.min_delta

    def __call__(self, epoch, metrics, logger=None):
        for m, v in metrics.items():
            if m in self.metric_history:
                self.metric_history[m].append(v)
        score = self._calculate_score(metrics)
        if logger:
            logger.log_scalar("early_stopping/score", score, epoch)
            logger.log_scalar("early_stopping/counter", self.counter, epoch)
        if epoch < self.min_epochs:
            return False
        if self.best_score is None or self._is_better(score, self.best_score):
            first = self.best_score is None
            self.best_score, self.best_metrics = score, metrics.copy()
            if not first:
                self.counter = 0
        else:
            self.counter += 1
            if self.verbose:
                print(f"EarlyStopping counter: {self.counter} out of {self.patience}")
            if self.counter >= self.patience:
                self.early_stop = True
                return True
        return False

    def get_best_metrics(self):
        return self.best_metrics

    def get_improvement_rate(self):
        return {m: (h[-1] - h[0]) / len(h) for m, h in self.metric_history.items() if len(h) > 1}


class SegmentationTrainer:
    def __init__(self, model, device):
        """model: segmentation model; device: device to train on."""
        self.model = model.to(device)
        self.device = device
        self.criterion = CrossEntropyLoss()
        self.logger = NullLogger(log_dir=Config.LOGS_DIR)
        self.num_classes = getattr(model, "classes", Config.NUM_CLASSES)
        self.log_metrics = True           # the reference computes metrics every batch; switch off for pure throughput
        self.grad_reducer = None          # ddp.GradAllReducer when data-parallel
        self.current_epoch = 0

    def calculate_metrics(self, outputs, masks):
        """Per-batch IoU / accuracy / per-class IoU (same keys as the reference)."""
        return segmentation_metrics(outputs, masks, self.num_classes)

    def train_step(self, images, masks, optimizer):
        """The timed hot path: reference train.py:340-344.  Returns (loss tensor, logits), no host sync."""
        optimizer.zero_grad()
        outputs = self.model(images)
        loss = self.criterion(outputs, masks)
        loss.backward()
        if self.grad_reducer is not None:
            self.grad_reducer.finish()
        optimizer.step()
        return loss, outputs

    def train_epoch(self, dataloader, optimizer, epoch):
        """Train for one epoch; returns the mean loss."""
        self.model.train()
        total_loss = 0.0
        for batch_idx, (images, masks) in enumerate(dataloader):
            images = images.to(self.device)
            masks = masks.to(self.device).long()
            loss, outputs = self.train_step(images, masks, optimizer)
            total_loss += loss.item()
            if self.log_metrics:
                with torch.no_grad():
                    metrics = self.calculate_metrics(outputs.detach(), masks)
                step = (epoch - 1) * len(dataloader) + batch_idx
                self.logger.log_scalar("train/loss", total_loss / (batch_idx + 1), step)
                self.logger.log_scalar("train/iou", metrics["iou"], step)
                self.logger.log_scalar("train/accuracy", metrics["accuracy"], step)
                self.logger.log_scalar("train/learning_rate", optimizer.param_groups[0]["lr"], step)
        return total_loss / len(dataloader)

    def validate(self, dataloader):
        """Validate the model; returns {'loss','iou','accuracy'}."""
        self.model.eval()
        total_loss = 0.0
        all_metrics = []
        with torch.no_grad():
            for batch_idx, (images, masks) in enumerate(dataloader):
                images = images.to(self.device)
                masks = masks.to(self.device).long()
                outputs = self.model(images)
                loss = self.criterion(outputs, masks)
                total_loss += loss.item()
                metrics = self.calculate_metrics(outputs, masks)
                all_metrics.append(metrics)
                if batch_idx % Config.LOG_INTERVAL == 0:
                    for c in range(self.num_classes):
                        self.logger.log_scalar(f"val/iou_class_{c}", metrics[f"iou_class_{c}"], self.current_epoch)
        avg = {"loss": total_loss / len(dataloader),
               "iou": float(np.mean([m["iou"] for m in all_metrics])),
               "accuracy": float(np.mean([m["accuracy"] for m in all_metrics]))}
        for k, v in avg.items():
            self.logger.log_scalar(f"val/{k}", v, self.current_epoch)
        return avg

    def train(self, train_dataloader, valid_dataloader, epochs, learning_rate, patience=7):
        """Train the model (Adam, early stopping on a weighted loss/IoU/accuracy score, best-checkpoint save)."""
        optimizer = FusedAdam(self.model.parameters(), lr=learning_rate)
        early_stopping = EarlyStopping(patience=patience, mode="max", min_epochs=10, metrics_to_track=["loss", "iou", "accuracy"],
                                       weights={"loss": -1.0, "iou": 1.0, "accuracy": 0.5}, verbose=True)
        self.current_epoch = 0
        for epoch in range(1, epochs + 1):
            self.current_epoch = epoch
            train_loss = self.train_epoch(train_dataloader, optimizer, epoch)
            valid_metrics = self.validate(valid_dataloader)
            print(f"Train Loss: {train_loss:.4f}")
            print(f"Valid Loss: {valid_metrics['loss']:.4f}")
            print(f"Valid Metrics: {valid_metrics}")
            if early_stopping(epoch, valid_metrics, self.logger):
                print(f"Early stopping triggered. Best metrics: {early_stopping.get_best_metrics()}")
                break
            if valid_metrics == early_stopping.get_best_metrics():
                os.makedirs(Config.CHECKPOINTS_DIR, exist_ok=True)
                torch.save({"epoch": epoch, "model_state_dict": self.model.state_dict(),
                            "optimizer_state_dict": optimizer.state_dict(), "metrics": valid_metrics,
                            "improvement_rates": early_stopping.get_improvement_rate()},
                           Path(Config.CHECKPOINTS_DIR) / "best_model.pth")
                print("Saved new best model!")
        self.logger.close()
