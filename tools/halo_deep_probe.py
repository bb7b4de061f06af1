#!/usr/bin/env python
"""Deep low-resolution bf16 3x3 layers: the shared implicit-GEMM source against the halo kernel's tile configurations
(UDASEG_HALO_CFG = 2: 8x32 px x 64 ch, 3: 8x32 x 128 ch, 6: 8x16 x 64 ch), forward, HIP-event time per launch.

    UDASEG_HALO_CFG=2 python tools/halo_deep_probe.py
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from uda_aerial_semantic_segmentation_research_amd import _lib, kernels as K  # noqa: E402

_lib.require_gpu()
K.ensure_workspace(torch.device("cuda", 0))
bf = torch.bfloat16
R = K.bn_replicas()


def t(fn, n=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for (n, h, w, ci, co) in [(8, 32, 32, 256, 256), (8, 16, 16, 512, 512), (8, 32, 32, 768, 256), (8, 48, 48, 256, 256), (8, 24, 24, 512, 512)]:
    d = K.conv_desc(n, h, w, ci, co, 3, 1, 1)
    x = torch.randn(n, h, w, ci, device="cuda").to(bf)
    wt = (torch.randn(co, 3, 3, ci, device="cuda") / math.sqrt(9 * ci)).to(bf)
    wf = torch.empty(K.frag_elems(co, ci, 3), device="cuda", dtype=bf)
    K.pack_frag_batched(wt, None, wf, torch.tensor([[0, 0, 0, co, ci, 3]], dtype=torch.int32, device="cuda"))
    y = torch.empty(n, h, w, co, device="cuda", dtype=bf)
    st = torch.zeros(R * 2 * co, dtype=torch.float64, device="cuda")
    a = t(lambda: K.conv2d_fwd_bf16(d, x, wt, None, None, y, 0, 0.0, st))
    b = t(lambda: K.conv2d_fwd_frag(d, x, None, wf, None, y, stats=st))
    fl = 2.0 * n * h * w * ci * co * 9
    print(f"cfg {os.environ.get('UDASEG_HALO_CFG', '-')} {n}x{h}x{w} {ci}->{co}: shared source {a:6.1f} us ({fl / a / 1e6:5.0f} TF)   halo {b:6.1f} us ({fl / b / 1e6:5.0f} TF)")
