"""Host-side pieces of the input pipeline (no GPU): the D4 composition table against the numpy oracle."""
import itertools

import numpy as np
import torch

from oracle.data_ref import geometric, normalize
from uda_aerial_semantic_segmentation_research_amd import data as D


def test_compose_d4_matches_numpy_ops():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (6, 6, 3), dtype=np.uint8)
    seen = set()
    for k, flip, tr in itertools.product(range(4), (None, 0, 1, -1), (False, True)):
        code = D.compose_d4(k, flip, tr)
        seen.add(code)
        want = geometric(img, k, flip, tr)
        got = np.empty_like(img)
        for y in range(6):
            for x in range(6):
                got[y, x] = img[D._apply_code(code, y, x, 6)]
        assert np.array_equal(got, want), (k, flip, tr, code)
    assert seen == set(range(8))


def test_normalize_constants():
    img = np.array([[[0, 128, 255]]], dtype=np.uint8)
    out = normalize(img)
    want = (np.array([0, 128, 255], dtype=np.float64) / 255 - np.array([0.485, 0.456, 0.406])) / np.array([0.229, 0.224, 0.225])
    assert np.allclose(out[0, 0], want, rtol=1e-6)
    codes = D.random_d4_codes(16, torch.Generator().manual_seed(3))
    assert codes.dtype == torch.int32 and codes.min() >= 0 and codes.max() <= 7
