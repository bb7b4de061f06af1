"""Debug aid: identity weights (center tap, o == c) and ramp input through the sixteen-wide-tile kernel: y[pix][ch] tells its source."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from uda_aerial_semantic_segmentation_research_amd import _lib, kernels as K
_lib.require_gpu()
K.ensure_workspace(torch.device("cuda", 0))
n, h, w, ci, co = 1, 8, 32, 16, 16
d = K.conv_desc(n, h, w, ci, co, 3, 1, 1)
for tap in (4, 0, 5, 8):
    wt = torch.zeros(co, 9, ci)
    for o in range(16):
        wt[o, tap, o] = 1.0
    x = (torch.arange(h * w).view(1, h, w, 1) * 16 + torch.arange(16).view(1, 1, 1, 16)).float() + 1.0
    packed = torch.zeros(K.n16_frag_elems(ci), device="cuda", dtype=torch.bfloat16)
    K.pack_up_batched(wt.view(co, 3, 3, ci).cuda(), None, packed, torch.tensor([[4, 0, 0, 16, ci, ci, 0, 0]], dtype=torch.int32, device="cuda"))
    y = torch.full((n, h, w, co), float("nan"), device="cuda")
    K.conv2d_fwd_n16(d, x.cuda(), packed, y)
    y = y.cpu()
    print(f"== tap {tap} (ky {tap // 3}, kx {tap % 3}): y[pix][ch] should be x[pix + (ky - 1, kx - 1)][ch] = 16 * srcpix + ch + 1")
    for (yy, xx) in [(0, 0), (0, 1), (1, 0), (3, 17), (7, 31)]:
        v = y[0, yy, xx]
        src = [(int((t - 1) // 16), int((t - 1) % 16)) if t > 0 else None for t in v.tolist()]
        print(f"   y[{yy},{xx}] = {[int(t) for t in v.tolist()]}  -> (srcpix, srcch) {src}   (this pixel's index {yy * w + xx})")
