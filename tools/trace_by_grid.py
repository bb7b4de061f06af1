#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV -> average duration per (kernel symbol, grid size): tells the launches of one kernel apart by
layer shape.     python tools/trace_by_grid.py <dir with *_kernel_trace.csv> [substring filter]"""
import csv
import glob
import re
import sys
from collections import defaultdict

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(list)
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("udaseg::", ""))
    if flt and flt not in name:
        continue
    grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    wg = int(r["Workgroup_Size_X"])
    acc[(name, grid // wg, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
rows = []
for (name, blocks, wg), v in acc.items():
    v.sort()
    rows.append((sum(v), name, blocks, wg, len(v), v[len(v) // 2], v[0]))
rows.sort(reverse=True)
for s, name, blocks, wg, n, med, mn in rows[:60]:
    print(f"{name[:66]:66s} blocks {blocks:6d} x{wg:4d}  n={n:4d}  median {med:8.1f} us  min {mn:8.1f}  total {s / 1e3:8.3f} ms")
