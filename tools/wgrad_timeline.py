#!/usr/bin/env python
"""Where a launch of the dominant weight-gradient kernel spends its cycles (conv_wgrad_h2_kernel, stamped twin): per wave role,
shader-clock cycles in {staging stores, barrier 1, MFMA phase, barrier 2} summed over the block's tiles, and the clock held.

    python tools/wgrad_timeline.py [n,h,w,ci,co ...]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from uda_aerial_semantic_segmentation_research_amd import _lib, kernels as K
    lib = _lib.require_gpu()
    K.ensure_workspace(torch.device("cuda", 0))
    shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]] or [(8, 128, 128, 64, 64), (8, 64, 64, 128, 128), (8, 32, 32, 256, 256)]
    cap = 1 << 14
    buf = torch.zeros(cap * 6, dtype=torch.int64, device="cuda")
    for (n, h, w, ci, co) in shapes:
        d = K.conv_desc(n, h, w, ci, co, 3, 1, 1)
        x = torch.randn(n, h, w, ci, device="cuda")
        dy = torch.randn(n, h, w, co, device="cuda")
        dw = torch.zeros(co, 3, 3, ci, device="cuda")
        for _ in range(5):
            K.conv2d_wgrad_halo(d, x, None, dy, dw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            K.conv2d_wgrad_halo(d, x, None, dy, dw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        buf.zero_()
        _lib.check(lib.udaseg_debug_set_timeline(buf.data_ptr(), cap))
        for _ in range(3):
            K.conv2d_wgrad_halo(d, x, None, dy, dw)
        torch.cuda.synchronize()
        _lib.check(lib.udaseg_debug_set_timeline(None, 0))
        t = buf.cpu().numpy().reshape(-1, 8)
        t = t[t[:, 7] == 1].astype(np.float64)
        fl = 2.0 * n * h * w * co * ci * 9
        print(f"== {(n, h, w, ci, co)}: back-to-back {us:.1f} us = {fl / us / 1e6:.1f} fp32-equivalent TFLOP/s; {len(t) // 2} blocks stamped")
        for role, name in ((0, "waves 0-3 (dx 0,1)"), (1, "waves 4-7 (dx 2 + staging)")):
            r = t[role::2]
            tiles = r[:, 6].mean()
            clk = (r[:, 4] / (r[:, 5] / 100.0)).mean() / 1e3      # cycles per microsecond / 1000 = GHz
            print(f"   {name:28s} tiles/block {tiles:5.1f}  clock {clk:.2f} GHz  whole role {r[:, 4].mean() / 1e3:7.1f} kcyc | per tile, cycles: "
                  f"staging {r[:, 0].mean() / tiles:7.0f}  barrier-1 wait {r[:, 1].mean() / tiles:7.0f}  MFMA phase {r[:, 2].mean() / tiles:7.0f}  "
                  f"barrier-2 wait {r[:, 3].mean() / tiles:7.0f}")


if __name__ == "__main__":
    main()
