"""Time the loss-family kernels (csrc/losses_seg.hip) at BASELINE's logits size and print achieved HBM GB/s against the
algorithmic bytes of each pass.  Usage (GPU box): python tools/bench_losses.py [--batch 8 --size 512 --classes 23]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--classes", type=int, default=23)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from uda_aerial_semantic_segmentation_research_amd import _lib, kernels as K
    _lib.require_gpu()
    n, c, hw = a.batch, a.classes, a.size * a.size
    ldc = (c + 3) // 4 * 4
    pixels = n * hw
    dev = "cuda"
    z1 = torch.randn(pixels, ldc, device=dev)
    z2 = torch.randn(pixels, ldc, device=dev)
    t = torch.randint(0, c, (pixels,), device=dev)
    w = torch.rand(c, device=dev) + 0.5
    d1, d2 = torch.empty_like(z1), torch.empty_like(z1)
    parts = torch.empty(K.seg_partials(), device=dev, dtype=torch.float64)
    sums = torch.zeros(n * 3 * c, device=dev, dtype=torch.float64)
    coef = torch.empty(n * 2 * c, device=dev)
    loss = torch.empty((), device=dev)
    lse = torch.empty(pixels, device=dev)
    ce_parts = torch.empty(_lib.load().udaseg_ce_partials(), device=dev, dtype=torch.float64)
    g = torch.ones((), device=dev)
    logit_b, tgt_b = pixels * ldc * 4, pixels * 8
    legs = {
        "ce_fwd (reference point)": (lambda: K.ce_fwd(z1, t, pixels, c, ldc, lse, ce_parts, loss), logit_b + tgt_b + pixels * 4),
        "ce_bwd (reference point)": (lambda: K.ce_bwd(z1, t, lse, g, pixels, c, ldc, d1), 2 * logit_b + tgt_b + pixels * 4),
        "dice_fwd": (lambda: K.dice_fwd(z1, t, n, hw, c, ldc, 1.0, sums, coef, loss), logit_b + tgt_b),
        "dice_bwd": (lambda: K.dice_bwd(z1, t, coef, g, 1.0, n, hw, c, ldc, d1), 2 * logit_b + tgt_b),
        "focal_fwd": (lambda: K.focal_fwd(z1, t, w, 0.25, 2.0, pixels, c, ldc, True, parts, loss), logit_b + tgt_b),
        "focal_bwd": (lambda: K.focal_bwd(z1, t, w, 0.25, 2.0, g, 1.0, pixels, c, ldc, d1), 2 * logit_b + tgt_b),
        "consistency_fwd": (lambda: K.consistency_fwd(z1, z2, 0.5, n, pixels, c, ldc, parts, loss), 2 * logit_b),
        "consistency_bwd": (lambda: K.consistency_bwd(z1, z2, 0.5, g, 1.0, n, pixels, c, ldc, d1, d2), 4 * logit_b),
    }
    out = {}
    for name, (fn, nbytes) in legs.items():
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        out[name] = {"us": round(us, 1), "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(nbytes / us / 1e3, 1)}
        print(f"{name:28s} {us:9.1f} us  {nbytes / 1e6:8.1f} MB  {nbytes / us / 1e3:8.1f} GB/s", flush=True)
    print(json.dumps({"workload": f"{n}x{c}x{a.size}x{a.size} logits", "legs": out}))


if __name__ == "__main__":
    main()
