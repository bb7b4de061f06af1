#!/usr/bin/env python
"""Where does a ~90 us implicit-GEMM launch lose its time?  Per-block timestamps from inside the kernel.

    python tools/igemm_timeline.py [n,h,w,ci,co,k,stride,pad ...]      (default: the 64-channel 128x128 layer and two more)

udaseg_debug_set_timeline makes every block of conv_igemm_kernel record {entry, first tile load, end of K loop, exit} in
100 MHz wall-clock ticks plus HW_ID / XCC_ID.  For one forward launch per shape this prints: the launch span, how the block
entries spread (dispatch ramp of the first round, start of the later rounds), the three phase durations, the idle time of
the 1024 resident block slots, and the spread of the exits (tail)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pct(a, q):
    return float(np.percentile(a, q))


def main():
    from uda_aerial_semantic_segmentation_research_amd import _lib, kernels as K
    lib = _lib.require_gpu()
    shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]] or [
        (8, 128, 128, 64, 64, 3, 1, 1), (8, 64, 64, 128, 128, 3, 1, 1), (8, 128, 128, 192, 64, 3, 1, 1), (32, 128, 128, 64, 64, 3, 1, 1)]
    cap = 1 << 16
    buf = torch.zeros(cap * 6, dtype=torch.int64, device="cuda")
    for (n, h, w, ci, co, k, s, p) in shapes:
        d = K.conv_desc(n, h, w, ci, co, k, s, p)
        dt = torch.bfloat16 if os.environ.get("TL_DTYPE") == "bf16" else torch.float32     # TL_DTYPE=bf16: the bf16 instantiation
        x = torch.randn(n, h, w, ci, device="cuda").to(dt)
        wt = (torch.randn(co, k, k, ci, device="cuda") * 0.05).to(dt)
        y = torch.empty(n, d.ho, d.wo, co, device="cuda", dtype=dt)
        for _ in range(5):
            K.conv2d_fwd(d, x, wt, None, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            K.conv2d_fwd(d, x, wt, None, y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        buf.zero_()
        _lib.check(lib.udaseg_debug_set_timeline(buf.data_ptr(), cap))
        K.conv2d_fwd(d, x, wt, None, y)
        torch.cuda.synchronize()
        _lib.check(lib.udaseg_debug_set_timeline(None, 0))
        t = buf.cpu().numpy().reshape(-1, 6)
        nb = int((t[:, 3] != 0).sum())
        t = t[:nb].astype(np.int64)
        t0 = t[:, 0].min()
        ent, ld, kend, ex = [(t[:, i] - t0) / 100.0 for i in range(4)]          # us
        span = ex.max()
        fl = 2.0 * n * d.ho * d.wo * co * ci * k * k
        print(f"== {(n, h, w, ci, co, k, s, p)}: {nb} blocks, back-to-back launch time {us:.1f} us ({fl / us / 1e6:.1f} TFLOP/s); "
              f"first entry -> last exit {span:.1f} us")
        order = np.argsort(ent)
        first = order[:min(nb, 1024)]
        print(f"   entries of the first {len(first)} blocks: p50 {pct(ent[first], 50):.1f}  p90 {pct(ent[first], 90):.1f}  "
              f"last {ent[first].max():.1f} us;  entries of the rest: first {ent[order[len(first):]].min() if nb > 1024 else float('nan'):.1f}  "
              f"p50 {pct(ent[order[len(first):]], 50) if nb > 1024 else float('nan'):.1f} us")
        pro, loop, epi = ld - ent, kend - ld, ex - kend
        for name, v in (("prologue (entry -> first load)", pro), ("K loop incl. pipeline fill", loop), ("epilogue", epi), ("whole block", ex - ent)):
            print(f"   {name:32s} p10 {pct(v, 10):6.1f}  p50 {pct(v, 50):6.1f}  p90 {pct(v, 90):6.1f}  max {v.max():6.1f} us")
        slots = min(nb, 1024)
        busy = (ex - ent).sum()
        print(f"   block-slot occupancy: sum of block lifetimes {busy:.0f} us over {slots} slots x {span:.1f} us = {100 * busy / (slots * span):.1f} %;  "
              f"K-loop share of the span: {100 * loop.sum() / (slots * span):.1f} %")
        print(f"   exits: p10 {pct(ex, 10):.1f}  p50 {pct(ex, 50):.1f}  p90 {pct(ex, 90):.1f}  p99 {pct(ex, 99):.1f}  last {span:.1f} us")
        xcc = t[:, 5] & 0xf
        print("   blocks per XCC: " + " ".join(str(int((xcc == i).sum())) for i in range(8)))
        print("   mean block lifetime per XCC (us): " + " ".join(f"{(ex - ent)[xcc == i].mean():.1f}" for i in range(8)))
        rnd1 = np.zeros(nb, dtype=bool)
        rnd1[first] = True
        for name, sel in (("first 1024 entries", rnd1), ("later entries", ~rnd1)):
            if sel.any():
                print(f"   {name:18s}: prologue p50 {pct(pro[sel], 50):5.1f} p90 {pct(pro[sel], 90):5.1f} | K loop p50 {pct(loop[sel], 50):5.1f} "
                      f"p90 {pct(loop[sel], 90):5.1f} | epilogue p50 {pct(epi[sel], 50):4.1f} p90 {pct(epi[sel], 90):4.1f} | exit p50 {pct(ex[sel], 50):6.1f} "
                      f"p99 {pct(ex[sel], 99):6.1f} us")
        # how many blocks are alive / inside their K loop as time goes by
        step = max(span / 24.0, 1.0)
        line_a, line_k = [], []
        for i in range(24):
            ts = (i + 0.5) * step
            line_a.append(int(((ent <= ts) & (ex > ts)).sum()))
            line_k.append(int(((ld <= ts) & (kend > ts)).sum()))
        print(f"   every {step:.1f} us: blocks alive   " + " ".join(f"{v:4d}" for v in line_a))
        print(f"   {'':13s} in the K loop " + " ".join(f"{v:4d}" for v in line_k))
        # gap between a slot's exit and the next entry on the same (XCC, CU/SE) -- HW_ID bits: cu 11:8, sh 12, se 15:13
        hw = t[:, 4]
        slot = (xcc.astype(np.int64) << 16) | (hw & 0xff00)
        gaps = []
        for sid in np.unique(slot):
            m = np.where(slot == sid)[0]
            e_sorted, x_sorted = np.sort(ent[m]), np.sort(ex[m])
            k = len(m) // 2 if len(m) >= 2 else 0
            # with c blocks resident per CU the (c + i)-th entry follows the i-th exit
            c = 4
            for i in range(max(0, len(m) - c)):
                gaps.append(e_sorted[c + i] - x_sorted[i])
        if gaps:
            gaps = np.array(gaps)
            print(f"   refill of a freed slot on the same CU (entry of block c+i minus i-th exit, c = 4): p10 {pct(gaps, 10):.1f} p50 {pct(gaps, 50):.1f} "
                  f"p90 {pct(gaps, 90):.1f} us over {len(gaps)} refills")


if __name__ == "__main__":
    main()
