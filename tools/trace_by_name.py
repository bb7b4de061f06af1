#!/usr/bin/env python
"""Neighbours (previous / next kernel on the same queue) of every launch whose symbol contains a substring, from a rocprofv3 kernel
trace directory:  python tools/trace_by_name.py <dir> copyBuffer"""
import collections
import csv
import glob
import sys

d, pat = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
pairs = collections.Counter()
for q, rs in byq.items():
    for i, r in enumerate(rs):
        if pat in r["Kernel_Name"]:
            p = rs[i - 1]["Kernel_Name"][:48] if i else "-"
            n = rs[i + 1]["Kernel_Name"][:48] if i + 1 < len(rs) else "-"
            pairs[(q, p, n, r["Grid_Size_X"])] += 1
for (q, p, n, g), c in pairs.most_common(40):
    print(f"{c:5d} x queue {q} grid {g:>8s} | after {p:48s} | before {n}")
