"""Stand-alone timing of the fp32-on-the-bf16-pipe forward kernels (csrc/conv_halo_f32x3.hip): one layer shape, one forced tile
configuration (udaseg_f32x3_force_config: 1-4 one-role kernels, 5-8 wave-specialised), HIP-event time per launch.

    python tools/f3_probe.py [n h w ci co cfg]

(The same script drove the timing-only probes of DESIGN section 5 -- a diagnostic build with uniform branches around the MFMAs,
the operand split, the fragment reads and the epilogue, selected by UDASEG_F3_PROBE; the branches cost 8 % and were removed.)
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from uda_aerial_semantic_segmentation_research_amd import _lib, kernels as K  # noqa: E402

_lib.require_gpu()
K.ensure_workspace(torch.device("cuda", 0))
n, h, w, ci, co, cfg = [int(v) for v in (sys.argv[1:7] if len(sys.argv) >= 7 else (8, 128, 128, 64, 64, 3))]
_lib.load().udaseg_f32x3_force_config(cfg)
g = torch.Generator().manual_seed(0)
x = torch.randn(n, h, w, ci, generator=g).cuda()
wt = (torch.randn(co, 3, 3, ci, generator=g) / math.sqrt(9 * ci)).cuda()
nf = 3 * K.frag_elems(co, ci, 3)
packed = torch.empty(nf, device="cuda", dtype=torch.bfloat16)
K.pack_frag_batched(wt, None, packed, torch.tensor([[0, 0, 0, co, ci, 3]], dtype=torch.int32, device="cuda"))
d = K.conv_desc(n, h, w, ci, co, 3, 1, 1)
y = torch.empty(n, h, w, co, device="cuda")
st = torch.zeros(K.bn_replicas() * 2 * co, dtype=torch.float64, device="cuda")
for _ in range(5):
    K.conv2d_fwd_frag(d, x, None, packed, None, y, stats=st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
R = 50
for _ in range(R):
    K.conv2d_fwd_frag(d, x, None, packed, None, y, stats=st)
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / R
fl = 2.0 * n * h * w * ci * co * 9
print(f"cfg {cfg} {n}x{h}x{w} {ci}->{co}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s fp32-equivalent")
