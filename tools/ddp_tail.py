#!/usr/bin/env python
"""End of the backward pass in a rocprofv3 kernel trace of bench.py: when the last weight-gradient kernel ends, which collective
kernels run (RCCL, queue of their own), when Adam starts -- the gap nothing can hide in a data-parallel step.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-also --no-sustain
    python tools/ddp_tail.py <kernel_trace.csv>
"""
import csv
import sys


def short(n):
    n = n.replace("void ", "").replace("udaseg::", "")
    return n[:n.index("(")] if "(" in n else n[:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_flat" in r["Kernel_Name"]]
    gaps, spans, busy = [], [], []
    for k in range(1, len(adam)):
        step = rows[adam[k - 1] + 1:adam[k] + 1]
        t0 = int(step[0]["Start_Timestamp"])
        wg = [r for r in step if "wgrad" in r["Kernel_Name"]]
        lastw = max(int(r["End_Timestamp"]) for r in wg)
        lastmain = max(int(r["End_Timestamp"]) for r in step[:-1] if "wgrad" not in r["Kernel_Name"] and r["Queue_Id"] == step[-1]["Queue_Id"])
        a0 = int(step[-1]["Start_Timestamp"])
        gaps.append(((a0 - lastw) / 1e3, (a0 - lastmain) / 1e3))
        spans.append((int(step[-1]["End_Timestamp"]) - t0) / 1e3)
    print(f"{len(gaps)} steps; step span (first kernel -> end of Adam) us: " + " ".join(f"{s:.0f}" for s in spans))
    print("last weight-gradient kernel end -> Adam start (us): " + " ".join(f"{g[0]:.0f}" for g in gaps))
    print("last main-queue kernel end      -> Adam start (us): " + " ".join(f"{g[1]:.0f}" for g in gaps))
    step = rows[adam[-2] + 1:adam[-1] + 1]
    t0 = int(step[0]["Start_Timestamp"])
    print("last step, its final kernels (start, end in us from the step's first kernel; queue; symbol):")
    for r in step[-10:]:
        print(f"  {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - t0) / 1e3:9.1f}  q{r['Queue_Id']}  {short(r['Kernel_Name']) or '(unnamed: RCCL)'}")
    anon = [r for r in step if not short(r["Kernel_Name"]) or "ccl" in r["Kernel_Name"].lower()]
    print(f"collective kernels in the last step: {len(anon)}; " + ", ".join(
        f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f} us x {r['Grid_Size_X']} threads" for r in anon))


if __name__ == "__main__":
    main()
