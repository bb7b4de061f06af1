"""Device-side input pipeline: decoded uint8 batches -> model input, on the GPU.

Upstream does this per sample on the host: ``cv2`` decode -> albumentations pipeline -> ``ToTensorV2``
(reference ``src/data/dataset.py:116-138``, ``src/models/augmentation.py:8-38``).  The parts with an exact definition
move to one HIP kernel (csrc/data_prep.hip): the D4 geometric augmentations (``RandomRotate90``, ``Flip``, ``Transpose``;
image and mask together), ``A.Normalize()`` and the layout change -- the output is already the channel-padded NHWC tensor
the stem convolution reads, handed to ``Unet`` as an ``[N,3,H,W]``-shaped view (no further copy).
The photometric / elastic augmentations of that pipeline (noise, blur, CLAHE, HSV, distortions) are albumentations
internals with no reference-side definition to match and stay on the host side of the boundary.
"""
import ctypes

import torch

from . import _lib
from ._lib import check
from ._operands import ops
from .engine import mark_padded_input

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # A.Normalize() defaults
IMAGENET_STD = (0.229, 0.224, 0.225)

# D4 code bits (see include/udaseg.h): out = fliph^b2(flipv^b1(transpose^b0(in)))
TRANSPOSE, FLIP_ROWS, FLIP_COLS = 1, 2, 4


def _apply_code(code, y, x, n):
    """Source coordinate read by output (y, x) of an n x n image under ``code``."""
    if code & FLIP_ROWS:
        y = n - 1 - y
    if code & FLIP_COLS:
        x = n - 1 - x
    return (x, y) if code & TRANSPOSE else (y, x)


def _code_of(fn):
    """The D4 code whose gather equals ``fn`` (a function on index grids), found on a 3 x 3 probe."""
    import numpy as np
    probe = np.arange(9).reshape(3, 3)
    want = fn(probe)
    for code in range(8):
        got = np.array([[probe[_apply_code(code, y, x, 3)] for x in range(3)] for y in range(3)])
        if np.array_equal(got, want):
            return code
    raise AssertionError("not a D4 element")


def compose_d4(rot90_k=0, flip=None, transpose=False):
    """D4 code of the basic pipeline's geometric steps in upstream's order (``augmentation.py:11-13``):
    ``np.rot90(img, rot90_k)`` (RandomRotate90), then ``cv2.flip(img, flip)`` for flip in {0: rows, 1: columns, -1: both}
    (Flip), then ``img.transpose(1, 0, 2)`` (Transpose)."""
    import numpy as np

    def fn(a):
        a = np.rot90(a, rot90_k % 4)
        if flip is not None:
            a = {0: a[::-1, :], 1: a[:, ::-1], -1: a[::-1, ::-1]}[flip]
        return a.T if transpose else a
    return _code_of(fn)


_COMPOSED = {}


def random_d4_codes(n, generator=None, p_rot90=0.5, p_flip=0.5, p_transpose=0.5):
    """One D4 code per sample, drawn with the basic training pipeline's branch probabilities (each step applied with
    p = 0.5; rotation factor uniform in {0..3}; flip code uniform in {-1, 0, 1}).  int32 CPU tensor."""
    g = generator
    u = torch.rand(n, 3, generator=g)
    k = torch.randint(0, 4, (n,), generator=g)
    d = torch.randint(-1, 2, (n,), generator=g)
    codes = []
    for i in range(n):
        key = (int(k[i]) if u[i, 0] < p_rot90 else 0, int(d[i]) if u[i, 1] < p_flip else None, bool(u[i, 2] < p_transpose))
        if key not in _COMPOSED:
            _COMPOSED[key] = compose_d4(*key)
        codes.append(_COMPOSED[key])
    return torch.tensor(codes, dtype=torch.int32)


def prepare_batch(images_u8, masks_u8=None, d4_codes=None, dtype=torch.float32, mean=IMAGENET_MEAN, std=IMAGENET_STD,
                  max_pixel_value=255.0):
    """images_u8 ``[N,H,W,3]`` uint8 (RGB, as decoded), masks_u8 ``[N,H,W]`` uint8 or None, d4_codes ``[N]`` int32 or None
    -> (images ``[N,3,H,W]``-shaped view of the padded NHWC buffer in ``dtype``, masks ``[N,H,W]`` int64 or None).
    Inputs may live on the host (moved with one async copy each) or on the GPU."""
    _lib.require_gpu()
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3:
        raise ValueError(f"prepare_batch: images must be uint8 [N,H,W,3], got {images_u8.dtype} {tuple(images_u8.shape)}")
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("prepare_batch: dtype must be torch.float32 or torch.bfloat16")
    n, h, w, _ = images_u8.shape
    if masks_u8 is not None and (masks_u8.dtype != torch.uint8 or tuple(masks_u8.shape) != (n, h, w)):
        raise ValueError(f"prepare_batch: masks must be uint8 [{n},{h},{w}], got {masks_u8.dtype} {tuple(masks_u8.shape)}")
    square_ok = 0
    if d4_codes is not None:
        if d4_codes.dtype != torch.int32 or tuple(d4_codes.shape) != (n,):
            raise ValueError("prepare_batch: d4_codes must be int32 [N]")
        if h != w:
            if d4_codes.device.type != "cpu":
                raise ValueError("prepare_batch: non-square images need host-side d4_codes (to rule out transposes)")
            if bool((d4_codes & TRANSPOSE).any()):
                raise ValueError("prepare_batch: transposing D4 codes need square images")
            square_ok = 1
    dev = torch.device("cuda", torch.cuda.current_device())
    img = images_u8.to(dev, non_blocking=True).contiguous()
    msk = None if masks_u8 is None else masks_u8.to(dev, non_blocking=True).contiguous()
    codes = None if d4_codes is None else d4_codes.to(dev, non_blocking=True).contiguous()
    cpad = 8 if dtype == torch.bfloat16 else 4
    out = torch.empty((n, h, w, cpad), device=dev, dtype=dtype)
    out_m = None if msk is None else torch.empty((n, h, w), device=dev, dtype=torch.int64)
    f3 = ctypes.c_float * 3
    # A.Normalize: mean*max_pixel_value and reciprocal(std*max_pixel_value), both rounded to fp32 first
    m255 = f3(*[float(torch.tensor(m, dtype=torch.float32) * max_pixel_value) for m in mean])
    r255 = f3(*[float(1.0 / (torch.tensor(s, dtype=torch.float32) * max_pixel_value)) for s in std])
    check(ops.udaseg_prepare_batch_u8(img, msk, codes, n, h, w, m255, r255, out, cpad, int(dtype == torch.bfloat16), out_m, square_ok,
                                      None), "prepare_batch_u8")
    mark_padded_input(out)
    return out.permute(0, 3, 1, 2)[:, :3], out_m


def synthetic_u8_batch(n, h, w, classes=23, seed=0, device="cuda"):
    """Seeded uint8 images / masks generated on the device (bench / smoke input: no dataset ships with the build)."""
    g = torch.Generator(device=device).manual_seed(seed)
    images = torch.randint(0, 256, (n, h, w, 3), generator=g, device=device, dtype=torch.uint8)
    masks = torch.randint(0, classes, (n, h, w), generator=g, device=device, dtype=torch.uint8)
    return images, masks
