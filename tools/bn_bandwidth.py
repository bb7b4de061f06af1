#!/usr/bin/env python
"""HBM rate of the BatchNorm passes at the shapes of the r18-Unet 8x3x512x512 step (fp32 or bf16: BN_DTYPE=bf16).

    python tools/bn_bandwidth.py

Each kernel is launched 20 times back to back on tensors of one layer's size (rotating over 4 buffer sets so a 4-16 MB layer
is not served from the Infinity Cache by its own previous launch); bytes = the passes the kernel makes over the activation
(bn_apply: read y, write z; bn_bwd_reduce: read dz, y; bn_bwd_apply: read dz, y, write dy), activation re-evaluated from y."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [(8 * 512 * 512, 16, 2), (8 * 256 * 256, 64, 1), (8 * 256 * 256, 32, 2), (8 * 128 * 128, 64, 6), (8 * 64 * 64, 128, 7),
          (8 * 32 * 32, 256, 7), (8 * 16 * 16, 512, 5)]          # (pixels, channels, BN layers of that size per step)


def main():
    from uda_aerial_semantic_segmentation_research_amd import _lib, kernels as K
    _lib.require_gpu()
    dt = torch.bfloat16 if os.environ.get("BN_DTYPE") == "bf16" else torch.float32
    R = K.bn_replicas()
    tot = {"bn_apply": 0.0, "bn_bwd_reduce": 0.0, "bn_bwd_apply": 0.0}
    print(f"dtype {dt}; us per launch and TB/s over the activation passes")
    for pixels, c, count in SHAPES:
        nset = 4
        ys = [torch.randn(pixels, c, device="cuda").to(dt) for _ in range(nset)]
        zs = [torch.empty_like(y) for y in ys]
        dzs = [torch.randn(pixels, c, device="cuda").to(dt) for _ in range(nset)]
        gamma, beta = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
        rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        sm, sr = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        sums = torch.zeros(R * 2 * c, dtype=torch.float64, device="cuda")
        sums.view(R, 2, c)[0, 0] = ys[0].double().sum(0)                      # what the producing conv's epilogue leaves
        sums.view(R, 2, c)[0, 1] = (ys[0].double() ** 2).sum(0)
        bs = torch.zeros(R * 2 * c, dtype=torch.float64, device="cuda")
        dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        es = ys[0].element_size()

        def run(fn, passes, name):
            for i in range(3):
                fn(i % nset)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(20):
                fn(i % nset)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            tot[name] += us * count
            return us, passes * pixels * c * es / us / 1e6
        a = run(lambda i: K.bn_apply(ys[i], sums, gamma, beta, None, zs[i], 1e-5, 0.1, rm, rv, sm, sr, 1, 0.0), 2, "bn_apply")
        if dt == torch.float32:
            r = run(lambda i: K.bn_bwd_reduce(dzs[i], None, ys[i], sm, sr, bs, 1, 0.0, gamma=gamma, beta=beta), 2, "bn_bwd_reduce")
            b = run(lambda i: K.bn_bwd_apply(dzs[i], None, ys[i], sm, sr, gamma, bs, zs[i], None, dg, db, 1, 0.0, beta=beta), 3,
                    "bn_bwd_apply")
        else:
            r = run(lambda i: K.bn_bwd_reduce(dzs[i], zs[i], ys[i], sm, sr, bs, 1, 0.0), 3, "bn_bwd_reduce")
            b = run(lambda i: K.bn_bwd_apply(dzs[i], zs[i], ys[i], sm, sr, gamma, bs, dzs[(i + 1) % nset], None, dg, db, 1, 0.0), 4,
                    "bn_bwd_apply")
        print(f"  {pixels:8d} x {c:4d} ({pixels * c * es / 1e6:6.1f} MB, x{count}): apply {a[0]:6.1f} us {a[1]:5.2f} | "
              f"bwd_reduce {r[0]:6.1f} us {r[1]:5.2f} | bwd_apply {b[0]:6.1f} us {b[1]:5.2f}")
    print("  per step (sum over layers): " + ", ".join(f"{k} {v / 1e3:.3f} ms" for k, v in tot.items()))


if __name__ == "__main__":
    main()
