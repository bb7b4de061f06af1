#!/usr/bin/env python
"""Every kernel of the last traced step of a rocprofv3 kernel trace of bench.py, in start order: start, end (us from the step's first
kernel), duration, queue, symbol.  python tools/step_timeline.py <kernel_trace.csv> [first [last]]"""
import csv
import sys


def short(n):
    n = n.replace("void ", "").replace("udaseg::", "")
    return n[:n.index("(")] if "(" in n else n[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_flat" in r["Kernel_Name"]]
    step = rows[adam[-2] + 1:adam[-1] + 1]
    t0 = int(step[0]["Start_Timestamp"])
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(step)
    qs = sorted({r["Queue_Id"] for r in step})
    print("# tools/step_timeline.py on a rocprofv3 --kernel-trace of bench.py (two streams): every kernel of the last traced step in start")
    print("# order -- index, start, end (us from the step's first kernel), duration, queue (q0 main, q1 side), symbol.  Durations inside")
    print("# the step are stretched by what runs beside them (the single-stream statistics are in the kernel_stats_serial file).")
    for i, r in enumerate(step[lo:hi], lo):
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{i:4d} {s:9.1f} {e:9.1f} {e - s:7.1f}  q{qs.index(r['Queue_Id'])}  {short(r['Kernel_Name'])}")


if __name__ == "__main__":
    main()
