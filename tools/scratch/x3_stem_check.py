import math, sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from uda_aerial_semantic_segmentation_research_amd import _lib, kernels as K
_lib.require_gpu(); K.ensure_workspace(torch.device("cuda", 0))
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().to("cuda", torch.float32)
def nchw(t): return t.detach().cpu().permute(0, 3, 1, 2).contiguous()
def e2(a, r): return ((a.double() - r).norm() / r.norm()).item()
for case in [(8, 512, 512, 4, 64, 7, 2, 3), (8, 128, 128, 64, 128, 3, 2, 1), (8, 128, 128, 64, 128, 1, 2, 0)]:
    n, h, w, ci, co, k, s, p = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, ci, h, w, generator=g)
    if ci == 4: x[:, 3] = 0
    wt = torch.randn(co, ci, k, k, generator=g) / math.sqrt(ci * k * k)
    d = K.conv_desc(n, h, w, ci, co, k, s, p)
    y_ref = F.conv2d(x.double(), wt.double(), None, s, p)
    dy = torch.randn(y_ref.shape, generator=g)
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), wt.shape, dy.double(), s, p)
    xd, wd, dyd = nhwc(x), wt.permute(0, 2, 3, 1).contiguous().cuda(), nhwc(dy)
    for mode in (1, 0):
        K.set_f32_split(mode)
        y = torch.empty((n, d.ho, d.wo, co), device="cuda")
        R = K.bn_replicas()
        st = torch.zeros(R * 2 * co, dtype=torch.float64, device="cuda")
        K.conv2d_fwd_bnstats(d, xd, wd, None, y, st)
        dw = torch.empty((co, k, k, ci), device="cuda")
        K.conv2d_wgrad(d, xd, dyd, dw)
        sums = st.view(R, 2, co).sum(0).cpu()
        yr = y_ref.permute(0, 2, 3, 1).reshape(-1, co)
        print(case, "mode", mode, "fwd l2", e2(nchw(y), y_ref), "wgrad l2", e2(dw.cpu().permute(0, 3, 1, 2), dw_ref),
              "stats sum err", ((sums[0] - yr.sum(0)).abs().max() / yr.sum(0).abs().max()).item(),
              "sq err", ((sums[1] - (yr * yr).sum(0)).abs().max() / (yr * yr).sum(0).abs().max()).item())
