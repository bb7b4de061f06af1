"""CPU oracle for the device-side input pipeline.

TEST INFRASTRUCTURE ONLY (see oracle/unet_ref.py header for who may import this).

numpy restatement of what the reference's dataset + basic augmentation pipeline do, for the parts with an exact definition:
* ``rotate90 / flip / transpose``  -- ``A.RandomRotate90`` = ``np.rot90(img, k)``, ``A.Flip`` = ``cv2.flip(img, d)``
  (d = 0 rows, 1 columns, -1 both), ``A.Transpose`` = ``img.transpose(1, 0, 2)`` (reference
  ``src/models/augmentation.py:11-13``), applied to image and mask alike;
* ``normalize``  -- ``A.Normalize()`` (``augmentation.py:36``): fp32 ``(img - mean*255) * reciprocal(std*255)``;
* ``to_model_input``  -- ``ToTensorV2`` + the dataset's mask cast (``src/data/dataset.py:131-136``): HWC -> CHW, mask int64.

PARITY UNPINNED for the library semantics: albumentations / cv2 are not installed here, so the three geometric ops and
Normalize are restated from their published definitions (they are plain index permutations and one affine map).
"""
import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def geometric(img, rot90_k=0, flip=None, transpose=False):
    a = np.rot90(img, rot90_k % 4)
    if flip is not None:
        a = {0: a[::-1], 1: a[:, ::-1], -1: a[::-1, ::-1]}[flip]
    if transpose:
        a = a.transpose(1, 0, 2) if a.ndim == 3 else a.T
    return np.ascontiguousarray(a)


def normalize(img_u8, max_pixel_value=255.0):
    mean = MEAN * np.float32(max_pixel_value)
    denom = np.reciprocal(STD * np.float32(max_pixel_value), dtype=np.float32)
    out = img_u8.astype(np.float32)
    out -= mean
    out *= denom
    return out


def to_model_input(img_u8, mask_u8, rot90_k=0, flip=None, transpose=False):
    """-> (float32 [3,H,W], int64 [H,W]) for one sample."""
    img = geometric(img_u8, rot90_k, flip, transpose)
    msk = geometric(mask_u8, rot90_k, flip, transpose)
    return normalize(img).transpose(2, 0, 1).copy(), msk.astype(np.int64)
