#!/usr/bin/env python
"""Host-side cost of the pieces of one kernel call (us per call, 2000 calls each, device idle between groups)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch


def t(fn, n=2000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return dt


def main():
    from uda_aerial_semantic_segmentation_research_amd import _lib, _operands, kernels as K
    _lib.require_gpu()
    K.ensure_workspace(torch.device("cuda", 0))
    lib = _lib.load()
    x = torch.randn(64, 64, device="cuda")
    y = torch.empty_like(x)
    st = K.stream()
    print(f"udaseg_version (ctypes, no args)        {t(lambda: lib.udaseg_version()):6.2f}")
    print(f"torch.empty((8,64,64,64))               {t(lambda: torch.empty((8, 64, 64, 64), device='cuda')):6.2f}")
    print(f"torch.empty_like                        {t(lambda: torch.empty_like(x)):6.2f}")
    print(f"K.conv_desc(...)                        {t(lambda: K.conv_desc(8, 64, 64, 64, 64, 3, 1, 1)):6.2f}")
    print(f"x.data_ptr()                            {t(lambda: x.data_ptr()):6.2f}")
    print(f"K.stream()                              {t(lambda: K.stream()):6.2f}")
    print(f"K.axpy (1 launch, checked wrapper)      {t(lambda: K.axpy(y, x, 1.0, st)):6.2f}")
    fn = _operands._FN["udaseg_axpy_f32"]
    yp, xp, n = y.data_ptr(), x.data_ptr(), x.numel()
    print(f"raw ctypes udaseg_axpy_f32              {t(lambda: fn(yp, xp, n, 1.0, st)):6.2f}")
    print(f"torch elementwise y.add_(x)             {t(lambda: y.add_(x)):6.2f}")
    n_, c = 8 * 32 * 32, 64
    yy = torch.randn(n_, c, device="cuda").bfloat16()
    zz = torch.empty_like(yy)
    R = K.bn_replicas()
    sums = torch.zeros(R * 2 * c, dtype=torch.float64, device="cuda")
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    sm, sr = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    import inspect
    print("bn_apply signature:", inspect.signature(K.bn_apply))
    print(f"K.bn_apply (bf16, 17 args)              {t(lambda: K.bn_apply(yy, sums, g, b, None, zz, 1e-5, 0.1, rm, rv, sm, sr, 1, 0.0, st)):6.2f}")
    ev = torch.cuda.Event()
    print(f"event.record + stream.wait_event        {t(lambda: (ev.record(), torch.cuda.current_stream().wait_event(ev))):6.2f}")


if __name__ == "__main__":
    main()
