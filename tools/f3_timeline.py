#!/usr/bin/env python
"""Where a launch of the wave-specialised split forward kernel (conv3x3_f32x3_ws_kernel, stamped twin) spends its cycles: per
block, one MFMA wave and one loader wave -- cycles until the first group is staged, cycles of work and of barrier waits inside
the K loop, epilogue, block lifetime, and the clock held.

    python tools/f3_timeline.py [n,h,w,ci,co ...]
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from uda_aerial_semantic_segmentation_research_amd import _lib, kernels as K
    lib = _lib.require_gpu()
    K.ensure_workspace(torch.device("cuda", 0))
    shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]] or [(8, 128, 128, 64, 64), (8, 64, 64, 128, 128), (8, 32, 32, 256, 256),
                                                                                   (8, 128, 128, 192, 64)]
    cap = 1 << 14
    buf = torch.zeros(cap * 6, dtype=torch.int64, device="cuda")
    R = K.bn_replicas()
    for (n, h, w, ci, co) in shapes:
        d = K.conv_desc(n, h, w, ci, co, 3, 1, 1)
        x = torch.randn(n, h, w, ci, device="cuda")
        wt = torch.randn(co, 3, 3, ci, device="cuda") / math.sqrt(9 * ci)
        nf = 3 * K.frag_elems(co, ci, 3)
        wf = torch.empty(nf, device="cuda", dtype=torch.bfloat16)
        K.pack_frag_batched(wt, None, wf, torch.tensor([[0, 0, 0, co, ci, 3]], dtype=torch.int32, device="cuda"))
        y = torch.empty(n, h, w, co, device="cuda")
        st = torch.zeros(R * 2 * co, dtype=torch.float64, device="cuda")
        for _ in range(5):
            K.conv2d_fwd_frag(d, x, None, wf, None, y, stats=st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            K.conv2d_fwd_frag(d, x, None, wf, None, y, stats=st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        buf.zero_()
        _lib.check(lib.udaseg_debug_set_timeline(buf.data_ptr(), cap))
        for _ in range(2):
            K.conv2d_fwd_frag(d, x, None, wf, None, y, stats=st)
        torch.cuda.synchronize()
        _lib.check(lib.udaseg_debug_set_timeline(None, 0))
        t = buf.cpu().numpy().reshape(-1, 8)
        t = t[t[:, 7] == 1].astype(np.float64)
        fl = 2.0 * n * h * w * co * ci * 9
        print(f"== {(n, h, w, ci, co)}: back-to-back {us:.1f} us = {fl / us / 1e6:.1f} fp32-equivalent TFLOP/s; {len(t) // 2} blocks stamped")
        if len(t) == 0:
            continue
        for role, name in ((0, "MFMA wave 0"), (1, "loader wave 4")):
            r = t[role::2]
            clk = (r[:, 4] / (r[:, 5] / 100.0)).mean() / 1e3
            ng = r[:, 6].mean()
            print(f"   {name:14s} clock {clk:.2f} GHz  lifetime {r[:, 4].mean() / 1e3:6.1f} kcyc = entry -> first group staged {r[:, 0].mean() / 1e3:5.1f}"
                  f" + K loop work {r[:, 1].mean() / 1e3:6.1f} + barrier waits {r[:, 2].mean() / 1e3:6.1f} ({ng:.0f} groups: {r[:, 1].mean() / ng:6.0f} + "
                  f"{r[:, 2].mean() / ng:6.0f} cycles per group) + epilogue {r[:, 3].mean() / 1e3:5.1f}")


if __name__ == "__main__":
    main()
