"""Checkpoint wire format: the two ``torch.save`` payloads upstream writes, so files move between the
reference and this build in either direction.

* training checkpoint, reference ``src/models/train.py:491-500``: ``{'epoch', 'model_state_dict',
  'optimizer_state_dict', 'metrics', 'improvement_rates'}`` at ``<dir>/best_model.pth``;
* phase checkpoint, reference ``src/models/phase_manager.py:75-151``: ``{'model_state_dict', 'metrics', 'phase',
  'timestamp'[, 'discriminator_state_dict']}`` at ``<dir>/best_model.pth`` or ``latest_model.pth``
  (phase names ``SEGMENTATION`` / ``ADVERSARIAL`` / ``FINE_TUNING``; the discriminator entry only for the last two).

``state_dict`` tensors are saved dense (logical OIHW shapes, smp / reference key names): the flat parameter arena and its
channel padding are an implementation detail that never reaches a file.
"""
import datetime
from pathlib import Path

import torch

PHASES = ("SEGMENTATION", "ADVERSARIAL", "FINE_TUNING")


def dense_state_dict(module):
    """``module.state_dict()`` with every tensor cloned to a dense CPU tensor (arena views carry their whole storage into
    ``torch.save`` otherwise)."""
    return {k: v.detach().to("cpu").contiguous().clone() for k, v in module.state_dict().items()}


def save_training_checkpoint(directory, epoch, model, optimizer, metrics, improvement_rates=None):
    path = Path(directory) / "best_model.pth"
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save({"epoch": epoch, "model_state_dict": dense_state_dict(model), "optimizer_state_dict": optimizer.state_dict(),
                "metrics": metrics, "improvement_rates": improvement_rates or {}}, path)
    return path


def save_phase_checkpoint(directory, model, metrics, phase, discriminator=None, is_best=False):
    if phase not in PHASES:
        raise ValueError(f"phase must be one of {PHASES}, got {phase!r}")
    payload = {"model_state_dict": dense_state_dict(model), "metrics": metrics, "phase": phase,
               "timestamp": datetime.datetime.now().isoformat()}
    if phase in PHASES[1:] and discriminator is not None:
        payload["discriminator_state_dict"] = dense_state_dict(discriminator)
    path = Path(directory) / ("best_model.pth" if is_best else "latest_model.pth")
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(payload, path)
    return path


def load_phase_checkpoint(directory, model, load_best=True, discriminator=None, map_location="cpu"):
    """Returns the checkpoint dict (model -- and discriminator, when given and present -- already loaded), or None when the
    file does not exist, like upstream."""
    path = Path(directory) / ("best_model.pth" if load_best else "latest_model.pth")
    if not path.exists():
        return None
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ckpt["model_state_dict"])
    if discriminator is not None and "discriminator_state_dict" in ckpt:
        discriminator.load_state_dict(ckpt["discriminator_state_dict"])
    return ckpt
