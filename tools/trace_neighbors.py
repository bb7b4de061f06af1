#!/usr/bin/env python
"""Which launches surround the runtime's own kernels in a rocprofv3 kernel trace?

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline
    python tools/trace_neighbors.py DIR [pattern ...]

Prints, for every kernel whose name matches a pattern (default: __amd_rocclr_copyBuffer, __amd_rocclr_fillBuffer,
at::native), the kernel that ran before and after it on the same queue in the LAST step of the trace -- enough to attribute
each stray copy / fill to the host code that issued it."""
import csv
import glob
import os
import sys
from collections import Counter


def main():
    d = sys.argv[1]
    pats = sys.argv[2:] or ["__amd_rocclr_copyBuffer", "__amd_rocclr_fillBuffer", "at::native"]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {d}")
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))

    def short(n):
        n = n.replace("void ", "").replace("udaseg::", "")
        return n[:n.index("(")] if "(" in n else n[:70]
    names = [short(r["Kernel_Name"]) for r in rows]
    # last step = from the last adam_flat_kernel but one to the last one
    adam = [i for i, n in enumerate(names) if n.startswith("adam_flat_kernel")]
    lo, hi = (adam[-2] + 1, adam[-1] + 1) if len(adam) >= 2 else (0, len(rows))
    ctx = Counter()
    for i in range(lo, hi):
        if any(p in rows[i]["Kernel_Name"] for p in pats):
            prev = names[i - 1] if i > lo else "-"
            nxt = names[i + 1] if i + 1 < hi else "-"
            dur = int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
            ctx[(names[i], prev, nxt, rows[i].get("Queue_Id", "?"))] += 1
            print(f"{i - lo:4d} q{rows[i].get('Queue_Id', '?'):>3} {dur:7d} ns  {names[i][:48]:48s} after {prev[:40]:40s} before {nxt[:40]}")
    # how busy was the device inside the step?  union of the kernel intervals over all queues vs the step's span
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows[lo:hi])
    span = max(e for _, e in iv) - iv[0][0]
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    gaps = []
    for st, en in iv[1:]:
        if st > cur_e:
            busy += cur_e - cur_s
            gaps.append(st - cur_e)
            cur_s, cur_e = st, en
        else:
            cur_e = max(cur_e, en)
    busy += cur_e - cur_s
    per_q = Counter()
    for r in rows[lo:hi]:
        per_q[r.get("Queue_Id", "?")] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    gaps.sort()
    print(f"\nlast step: span {span / 1e6:.3f} ms, some kernel running {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %), "
          f"{len(gaps)} idle gaps totalling {sum(gaps) / 1e6:.3f} ms (median {gaps[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us, "
          f"max {gaps[-1] / 1e3 if gaps else 0:.1f} us); kernel time per queue: "
          + ", ".join(f"q{q}: {t / 1e6:.3f} ms" for q, t in per_q.items()))
    print(f"last step: {hi - lo} launches; matched {sum(ctx.values())}")
    for (n, prev, nxt, q), c in ctx.most_common():
        print(f"{c:4d} x {n[:40]:40s} | after {prev[:36]:36s} | before {nxt[:36]}")


if __name__ == "__main__":
    main()
