"""Stand-alone timing of the phase kernels (csrc/conv_up_f32x3.hip) on the decoder conv1 shapes of BASELINE cfg 2, one forced tile
configuration after the other (udaseg_up_f32x3_force_config 1..8), next to the nine-tap kernel over the same virtual input.

    python tools/up_probe.py [cfgs...]         # HIP-event time per launch, back to back
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from uda_aerial_semantic_segmentation_research_amd import _lib, kernels as K  # noqa: E402

_lib.require_gpu()
lib = _lib.load()
K.ensure_workspace(torch.device("cuda", 0))
bf = torch.bfloat16
# (n, output h, w, up-sampled channels, skip channels, produced channels): r18-Unet decoder blocks 0..4 at 8 x 512^2
SHAPES = [(8, 32, 32, 512, 256, 256), (8, 64, 64, 256, 128, 128), (8, 128, 128, 128, 64, 64), (8, 256, 256, 64, 64, 32),
          (8, 512, 512, 32, 0, 16)]
cfgs = [int(v) for v in sys.argv[1:]] or list(range(1, 9))


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for n, h, w, ca, cs, co in SHAPES:
    g = torch.Generator().manual_seed(0)
    ci = ca + cs
    a = torch.randn(n, h // 2, w // 2, ca, generator=g).cuda()
    dy = torch.randn(n, h, w, co, generator=g).cuda()
    w32 = (torch.randn(co, 3, 3, ci, generator=g) / math.sqrt(9 * ci)).cuda()
    wt32 = w32.permute(3, 1, 2, 0).contiguous()
    n_uf, n_ub = 3 * K.frag_elems(co, ca, 4), 3 * K.frag_elems(ca, co, 4)
    packed = torch.empty(n_uf + n_ub, device="cuda", dtype=bf)
    K.pack_up_batched(w32, wt32, packed, torch.tensor([[2, 0, 0, co, ca, ci, 0, 0], [3, 0, n_uf, ca, co, co, 0, 0]],
                                                       dtype=torch.int32, device="cuda"))
    d = K.conv_desc(n, h, w, ci, co, 3, 1, 1)
    y = torch.zeros(n, h, w, co, device="cuda")
    da = torch.empty(n, h // 2, w // 2, ca, device="cuda")
    st = torch.zeros(K.bn_replicas() * 2 * co, dtype=torch.float64, device="cuda")
    gf = 2.0 * n * h * w * ca * co * 9 / 1e9            # the nine-tap FLOPs of the up-sampled half
    # nine-tap reference: the up-sampled half alone through the fused gather (up_ca == ci, no skip)
    d9 = K.conv_desc(n, h, w, ca, co, 3, 1, 1)
    w9 = w32[..., :ca].contiguous()
    p9 = torch.empty(3 * K.frag_elems(co, ca, 3), device="cuda", dtype=bf)
    K.pack_frag_batched(w9, None, p9, torch.tensor([[0, 0, 0, co, ca, 3]], dtype=torch.int32, device="cuda"))
    us9 = timed(lambda: K.conv2d_fwd_frag(d9, a, None, p9, None, y, stats=st, up=True))
    print(f"== {n}x{h}x{w} up {ca} (+ skip {cs}) -> {co}: nine-tap gather forward {us9:7.1f} us ({gf / us9 * 1e3:6.1f} TF/s nine-tap-equivalent)")
    for cfg in cfgs:
        lib.udaseg_up_f32x3_force_config(cfg)
        uf = timed(lambda: K.conv2d_fwd_up(d, a, packed[:n_uf], y, accumulate=cs > 0, stats=st))
        ub = timed(lambda: K.conv2d_dgrad_up(d, dy, ca, packed[n_uf:], da))
        print(f"   cfg {cfg}: forward {uf:7.1f} us ({gf / uf * 1e3:6.1f} TF/s nine-tap-equivalent)   data gradient {ub:7.1f} us ({gf / ub * 1e3:6.1f})")
    lib.udaseg_up_f32x3_force_config(0)
    uf = timed(lambda: K.conv2d_fwd_up(d, a, packed[:n_uf], y, accumulate=cs > 0, stats=st))
    ub = timed(lambda: K.conv2d_dgrad_up(d, dy, ca, packed[n_uf:], da))
    print(f"   heuristic: forward {uf:7.1f} us   data gradient {ub:7.1f} us")
