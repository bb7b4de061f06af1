"""The CPU restatement of the reference's Dice / focal / consistency / fine-tuning losses (oracle/losses_ref.py) against
the fixture the REFERENCE's own classes produced (tests/golden/losses_ref.npz, made by oracle/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import losses_ref as O

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses_ref.npz"))
CASES = ("c23", "c5", "c2")


def case_inputs(name, dtype):
    seed, batch, classes, h, w = (int(v) for v in GOLD[f"{name}/shape"])
    return classes, O.loss_inputs(seed, batch, classes, h, w, dtype=dtype)


def oracle_fns(classes, target, weights):
    return {
        "dice": lambda z: O.DiceLossRef()(z, target),
        "dice_smooth": lambda z: O.DiceLossRef(0.1)(z, target),
        "focal": lambda z: O.WeightedSegmentationLossRef(classes, weights).focal_loss(z, target),
        "wseg": lambda z: O.WeightedSegmentationLossRef(classes, weights)(z, target, 0.7),
        "wseg_sum": lambda z: O.WeightedSegmentationLossRef(classes, None, 0.5, 1.5, 'sum')(z, target),
        "cons": lambda a, b: O.ConsistencyLossRef()(a, b),
        "cons_t2": lambda a, b: O.ConsistencyLossRef(2.0)(a, b),
        "fine": lambda a, b, d, s: O.FineTuningLossRef()(a, b, d, 10, s, target)['total'],
    }


def inputs_of(key, z1, z2, domain):
    return {"cons": (z1, z2), "cons_t2": (z1, z2), "fine": (z1, z2, domain, z2)}.get(key, (z1,))


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("name", CASES)
def test_restatement_matches_reference_f64(name):
    classes, (z1, z2, target, weights, domain) = case_inputs(name, torch.float64)
    for key, fn in oracle_fns(classes, target, weights).items():
        leaves = [t.clone().requires_grad_(True) for t in inputs_of(key, z1, z2, domain)]
        val = fn(*leaves)
        val.backward()
        assert rel(val.item(), GOLD[f"{name}/f64/{key}/value"]) < 1e-12, key
        for i, t in enumerate(leaves):
            assert rel(t.grad.numpy(), GOLD[f"{name}/f64/{key}/grad{i}"]) < 1e-12, (key, i)


@pytest.mark.parametrize("name", CASES)
def test_restatement_matches_reference_f32_values(name):
    classes, (z1, z2, target, weights, domain) = case_inputs(name, torch.float32)
    for key, fn in oracle_fns(classes, target, weights).items():
        val = fn(*inputs_of(key, z1, z2, domain))
        assert rel(val.item(), GOLD[f"{name}/f32/{key}/value"]) < 2e-5, key
        # and the reference's own fp32 result sits within fp32 rounding of its fp64 result
        assert rel(GOLD[f"{name}/f32/{key}/value"], GOLD[f"{name}/f64/{key}/value"]) < 2e-5, key


@pytest.mark.parametrize("name", CASES)
def test_fine_tuning_dict_and_rampup(name):
    classes, (z1, z2, target, weights, domain) = case_inputs(name, torch.float64)
    ft = O.FineTuningLossRef(0.8, 0.2, 0.3, rampup_length=8)
    assert [ft.rampup(e) for e in (0, 3, 8, 50)] == [0.0, 0.375, 1.0, 1.0]
    keys = list(GOLD[f"{name}/fine_dict/keys"])
    assert keys == ['consistency', 'domain_confusion', 'rampup_weight', 'supervised', 'total']
    for epoch in (0, 3, 8, 50):
        d = ft(z1, z2, domain, epoch, z2, target.float())
        assert rel([d[k].item() for k in keys], GOLD[f"{name}/fine_dict/{epoch}"]) < 1e-12
    d = ft(z1, z2, domain, 3)
    assert d['supervised'].item() == 0.0
    assert rel([d[k].item() for k in keys], GOLD[f"{name}/fine_dict/unsup"]) < 1e-12


def test_class_weights():
    g = torch.Generator().manual_seed(5)
    ds = [(None, torch.randint(0, 6, (9, 11), generator=g)) for _ in range(4)]
    for method in ("effective_samples", "inverse_freq"):
        w = O.calculate_class_weights_ref(ds, 7, method)
        assert np.array_equal(w.numpy(), GOLD[f"class_weights/{method}"])
        assert abs(w.sum().item() - 7.0) < 1e-5


def test_package_class_weights_matches_fixture():
    # host-side helper of the product (pure torch, no GPU): same numbers as the reference's
    from uda_aerial_semantic_segmentation_research_amd.losses import calculate_class_weights
    g = torch.Generator().manual_seed(5)
    ds = [(None, torch.randint(0, 6, (9, 11), generator=g)) for _ in range(4)]
    for method in ("effective_samples", "inverse_freq"):
        assert np.allclose(calculate_class_weights(ds, 7, method).numpy(), GOLD[f"class_weights/{method}"], rtol=1e-6, atol=0)


def test_dice_accepts_one_hot_targets():
    classes, (z1, _, target, _, _) = case_inputs("c5", torch.float64)
    oh = torch.nn.functional.one_hot(target, classes).permute(0, 3, 1, 2).double()
    assert O.DiceLossRef()(z1, oh).item() == O.DiceLossRef()(z1, target).item()
