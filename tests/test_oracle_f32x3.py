"""CPU: the three-term bf16 split that the fp32 convolutions run on (oracle/f32x3_ref.py restates csrc/halo_common.h::split3).

Pins the two facts the "fp32 grade" claim of DESIGN section 3 rests on, without a GPU: the split is EXACT, and the six products
with i + j <= 2 miss the fp32 product by at most one unit in its last place."""
import numpy as np

from oracle.f32x3_ref import bf16_round, six_products, split3


def _samples(n, seed, lo=-60, hi=60):
    g = np.random.default_rng(seed)
    return (g.standard_normal(n) * np.exp2(g.integers(lo, hi, n))).astype(np.float32)


def test_bf16_round_is_round_to_nearest_even():
    import torch
    x = _samples(200000, 0)
    assert np.array_equal(bf16_round(x), torch.from_numpy(x).to(torch.bfloat16).float().numpy())


def test_split_is_exact_over_120_binades():
    x = np.concatenate([_samples(400000, 1), np.float32([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.0e-30, 2.0 ** -100])])
    t0, t1, t2 = split3(x)
    assert np.array_equal(t0.astype(np.float64) + t1.astype(np.float64) + t2.astype(np.float64), x.astype(np.float64))
    for t in (t0, t1, t2):                                     # every term IS a bf16 number
        assert np.array_equal(bf16_round(t), t)
    a = np.abs(x.astype(np.float64))
    assert (np.abs(t1) <= a * 2.0 ** -8).all() and (np.abs(t2) <= a * 2.0 ** -16).all()


def test_six_products_miss_the_product_by_at_most_one_fp32_ulp():
    a, b = _samples(300000, 2, -20, 20), _samples(300000, 3, -20, 20)
    exact = a.astype(np.float64) * b.astype(np.float64)
    err = np.abs(six_products(a, b) - exact)
    assert (err <= np.abs(exact) * 2.0 ** -23 * (1 + 2.0 ** -8)).all()
    # typical size: far below the bound (the left-out terms have random signs)
    assert np.median(err / np.abs(exact)) <= 2.0 ** -26
    # a dot product of 4608 terms (a 512-channel 3x3 layer): the split adds less than the fp32 summation's own rounding
    g = np.random.default_rng(4)
    x, w = g.standard_normal((64, 4608)).astype(np.float32), g.standard_normal((64, 4608)).astype(np.float32)
    ref = (x.astype(np.float64) * w.astype(np.float64)).sum(1)
    e_split = np.abs(six_products(x, w).sum(1) - ref)
    e_fp32 = np.abs(np.cumsum((x * w).astype(np.float32), axis=1, dtype=np.float32)[:, -1].astype(np.float64) - ref)
    assert e_split.max() <= 0.05 * e_fp32.max() + 1e-9


def test_sign_pattern_of_the_k_loop():
    """oracle/f32x3_ref.py::negated_groups restates csrc/halo_common.h::f3_negated_groups: the (chunk, dx) groups [q1, q3) of a 3x3
    layer's K loop that run on negated weights and a negated accumulator (+ - - +).  The middle block is never empty, never the
    whole loop, and centred -- what makes the truncation bias of the two halves cancel."""
    from oracle.f32x3_ref import negated_groups
    for nk16 in range(1, 65):
        g = negated_groups(nk16)
        ng = 3 * nk16
        assert 0 < g.start < g.stop < ng
        assert g.start == ng - g.stop                       # as many positive groups in front as behind
        assert abs(len(g) - ng / 2) <= 1.0                  # about half of the loop
    assert list(negated_groups(1)) == [1] and list(negated_groups(2)) == [2, 3] and list(negated_groups(4)) == [3, 4, 5, 6, 7, 8]
    # and in exact arithmetic the pattern changes nothing: sum of +g over the outer groups minus (-g) over the middle ones
    rng = np.random.default_rng(0)
    terms = rng.standard_normal(3 * 8)
    mid = negated_groups(8)
    acc = 0.0
    for G, t in enumerate(terms):
        if G == mid.start or G == mid.stop:
            acc = -acc
        acc += -t if G in mid else t
    assert abs(acc - terms.sum()) < 1e-12
