"""Time single convolution launches (forward / dgrad / wgrad) through the C-ABI: python tools/bench_conv.py [shapes...]
shape = n,h,w,ci,co,k,stride,pad   (default: a K sweep of the 64-channel 128x128 layer)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def time_it(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    from uda_aerial_semantic_segmentation_research_amd import _lib, kernels as K
    _lib.require_gpu()
    K.ensure_workspace(torch.device("cuda"))
    shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]] or [
        (8, 128, 128, 64, 64, 1, 1, 0), (8, 128, 128, 64, 64, 3, 1, 1), (8, 128, 128, 64, 64, 5, 1, 2),
        (8, 128, 128, 64, 64, 7, 1, 3), (8, 64, 64, 128, 128, 3, 1, 1), (8, 32, 32, 256, 256, 3, 1, 1),
        (8, 16, 16, 512, 512, 3, 1, 1), (8, 256, 256, 128, 32, 3, 1, 1), (8, 256, 256, 64, 128, 4, 2, 1)]
    R = K.bn_replicas()
    print(f"{'shape':34s} {'fwd us':>9s} {'TF':>6s} {'fwd+stats':>9s} {'TF':>6s} {'dgrad us':>9s} {'TF':>6s} {'wgrad us':>9s} {'TF':>6s}")
    for (n, h, w, ci, co, k, s, p) in shapes:
        d = K.conv_desc(n, h, w, ci, co, k, s, p)
        x = torch.randn(n, h, w, ci, device="cuda")
        wt = torch.randn(co, k, k, ci, device="cuda") * 0.05
        y = torch.empty(n, d.ho, d.wo, co, device="cuda")
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.zeros_like(wt)
        stats = torch.zeros(R * 2 * co, dtype=torch.float64, device="cuda")
        wt_t = torch.empty(ci * k * k * co, device="cuda")
        K.pack_dgrad_weights(d, wt, wt_t)
        fl = 2.0 * n * d.ho * d.wo * co * ci * k * k
        only = os.environ.get("BENCH_CONV_ONLY", "")        # fwd | dgrad | wgrad: profile one kind (others print 0)
        skip = lambda kind: only and only != kind
        t_f = 1e-9 if skip("fwd") else time_it(lambda: K.conv2d_fwd(d, x, wt, None, y))
        t_s = 1e-9 if only else time_it(lambda: K.conv2d_fwd_bnstats(d, x, wt, None, y, stats))
        t_d = 1e-9 if skip("dgrad") else time_it(lambda: K.conv2d_dgrad(d, dy, wt_t, dx))
        t_w = 1e-9 if skip("wgrad") else time_it(lambda: K.conv2d_wgrad(d, x, dy, dw, False))
        tf = lambda t: fl / t / 1e6
        print(f"{str((n, h, w, ci, co, k, s, p)):34s} {t_f:9.1f} {tf(t_f):6.1f} {t_s:9.1f} {tf(t_s):6.1f} {t_d:9.1f} {tf(t_d):6.1f} "
              f"{t_w:9.1f} {tf(t_w):6.1f}", flush=True)


if __name__ == "__main__":
    main()
