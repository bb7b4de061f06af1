import math, sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from uda_aerial_semantic_segmentation_research_amd import _lib, kernels as K
_lib.require_gpu(); K.ensure_workspace(torch.device("cuda", 0))
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().to("cuda", torch.float32)
def nchw(t): return t.detach().cpu().permute(0, 3, 1, 2).contiguous()
def rep(tag, y, r):
    e = y.double() - r
    m = r.abs().mean()
    print(f"{tag:34s} l2 {(e.norm() / r.norm()).item():.3e}  mean signed err / mean|y| {(e.mean() / m).item():+.3e}  toward-zero bias {((e * r.sign()).mean() / m).item():+.3e}")
for case in [(8, 512, 512, 4, 64, 7, 2, 3), (8, 128, 128, 64, 64, 3, 1, 1), (8, 128, 128, 64, 128, 3, 2, 1)]:
    n, h, w, ci, co, k, s, p = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)
    d = K.conv_desc(n, h, w, ci, co, k, s, p)
    y_ref = F.conv2d(x.double(), wt.double(), None, s, p)
    xd, wd = nhwc(x), wt.permute(0, 2, 3, 1).contiguous().cuda()
    for mode in (1, 0):
        K.set_f32_split(mode)
        y = torch.empty((n, d.ho, d.wo, co), device="cuda")
        K.conv2d_fwd(d, xd, wd, None, y)
        rep(f"{case} shared mode {mode}", nchw(y), y_ref)
    if k == 3 and s == 1:
        wf = torch.empty(3 * K.frag_elems(co, ci, 3), device="cuda", dtype=torch.bfloat16)
        K.pack_frag_batched(wd, None, wf, torch.tensor([[0, 0, 0, co, ci, 3]], dtype=torch.int32, device="cuda"))
        y = torch.empty((n, h, w, co), device="cuda")
        K.conv2d_fwd_frag(d, xd, None, wf, None, y)
        rep(f"{case} halo f32x3", nchw(y), y_ref)
