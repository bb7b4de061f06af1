"""``Config`` -- the class-attribute singleton the reference imports from ``src/models/config.py``.  That file is
git-ignored upstream (reference ``.gitignore:50-51``), so the attributes below are reconstructed from their uses
(SURVEY Appendix A): the traced model is Unet/resnet50, 23 classes, 3 input channels, 256x256.
"""
import os

import torch


class Config:
    MODEL_NAME = "Unet"
    ENCODER_NAME = "resnet50"
    ENCODER_WEIGHTS = None            # 'imagenet' upstream; no downloads offline
    IN_CHANNELS = 3
    NUM_CLASSES = 23
    IMAGE_SIZE = (256, 256)
    NORMALIZE_MEAN = (0.485, 0.456, 0.406)
    NORMALIZE_STD = (0.229, 0.224, 0.225)
    NUM_EPOCHS = 100
    BATCH_SIZE = 4
    LEARNING_RATE = 1e-4
    PATIENCE = 7
    LOG_INTERVAL = 10
    TRAIN_VAL_SPLIT = 0.8
    NUM_WORKERS = 0
    DATA_DIR = "data"
    SAMPLE_DATA_DIR = "data/sample/semantic_drone"
    LOGS_DIR = "logs"
    CHECKPOINTS_DIR = "checkpoints"
    CHECKPOINT_DIR = "checkpoints"    # the singular spelling is also used upstream (train.py:674,680)
    DEVICE = "cuda" if torch.cuda.is_available() else "cpu"

    @classmethod
    def get_device(cls):
        return torch.device("cuda" if torch.cuda.is_available() else "cpu")

    @classmethod
    def setup_directories(cls):
        for d in (cls.LOGS_DIR, cls.CHECKPOINTS_DIR):
            os.makedirs(d, exist_ok=True)
