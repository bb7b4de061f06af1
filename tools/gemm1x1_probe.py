#!/usr/bin/env python
"""Stand-alone timing of the bf16 1x1 / stride-1 convolution kernels on r50's bottleneck shapes (forward with fused BatchNorm
statistics, data gradient): HIP-event time per launch, back to back.  UDASEG_GEMM_1X1=0 keeps the streaming kernel everywhere.

    python tools/gemm1x1_probe.py [M,K,N ...]
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
shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]] or [
    (294912, 64, 256), (294912, 256, 64), (73728, 128, 512), (73728, 512, 128), (18432, 256, 1024), (18432, 1024, 256),
    (4608, 512, 2048), (4608, 2048, 512)]
for (M, ci, co) in shapes:
    n, h, w = 8, int(math.isqrt(M // 8)), int(math.isqrt(M // 8))
    assert n * h * w == M
    x = torch.randn(n, h, w, ci, device="cuda").to(bf)
    wt = (torch.randn(co, ci, device="cuda") / math.sqrt(ci)).to(bf)
    nf = K.frag_elems(co, ci, 1)
    wf = torch.empty(nf, device="cuda", dtype=bf)
    K.pack_frag_batched(wt.contiguous(), None, wf, torch.tensor([[0, 0, 0, co, ci, 1]], dtype=torch.int32, device="cuda"))
    d = K.conv_desc(n, h, w, ci, co, 1, 1, 0)
    y = torch.empty(n, h, w, co, device="cuda", dtype=bf)
    st = torch.zeros(K.bn_replicas() * 2 * co, dtype=torch.float64, device="cuda")
    for _ in range(5):
        K.conv2d_fwd_frag(d, x, None, wf, None, y, stats=st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    R = 50
    for _ in range(R):
        K.conv2d_fwd_frag(d, x, None, wf, None, y, stats=st)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / R
    fl = 2.0 * M * ci * co
    by = (M * (ci + co) + ci * co) * 2
    print(f"M={M:6d} K={ci:4d} N={co:4d}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  {by / us / 1e6:5.2f} TB/s algorithmic")
