"""Launch floor of the small BatchNorm passes: bn_apply (training form: 16 replica sums per channel in the preamble), bn_apply_eval (light
preamble) and a plain fill on the same tensors, back to back.  Round 5: 11.6 / 10.0 / 3.9 us at 8 MB, 14.6 / 12.7 / 6.3 us at 67 MB --
the ~6 us above a fill are not the dependent round trip of the preamble (requesting the first stream batch before it: 8 MB unchanged,
67 MB 14.5 -> 24.7 us from the registers it holds, step -3.6 %; not kept).   python tools/bn_floor.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from uda_aerial_semantic_segmentation_research_amd import kernels as K
def t(f, it=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
R = K.bn_replicas()
for (n, h, w, c) in [(8, 16, 16, 512), (8, 32, 32, 256), (8, 64, 64, 128), (8, 128, 128, 64)]:
    y = torch.randn(n, h, w, c, device="cuda"); z = torch.empty_like(y)
    sums = torch.zeros(R * 2 * c, dtype=torch.float64, device="cuda")
    K.bn_stats(y, sums) if hasattr(K, "bn_stats") else None
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    sm, sr = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    import inspect
    ta = t(lambda: K.bn_apply(y, sums, g, b, None, z, 1e-5, 0.1, rm, rv, sm, sr, 1, 0.0))
    te = t(lambda: K.bn_apply_eval(y, g, b, rm, rv, None, z, 1e-5, 1, 0.0))
    tf = t(lambda: z.zero_())
    print(f"{(n,h,w,c)}: bn_apply (train) {ta:6.1f} us   bn_apply_eval {te:6.1f} us   fill {tf:6.1f} us   bytes {2*y.numel()*4/1e6:.1f} MB")
