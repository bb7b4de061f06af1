#!/usr/bin/env python
"""Compare two `bench.py --layer-table` stderr dumps layer by layer:  tools/layer_diff.py OLD.err NEW.err [min_ms]"""
import re
import sys


def parse(f):
    rows = {}
    for l in open(f):
        m = re.match(r'(fwd|dgrad|wgrad)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+) \|\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)', l)
        if m:
            rows[tuple(m.group(i) for i in range(1, 9))] = tuple(float(m.group(i)) for i in range(9, 13))
    return rows


a, b = parse(sys.argv[1]), parse(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
print('kind n hi wi ci co k s | calls  old_ms new_ms  GFLOP  oldTF newTF')
ta = tb = 0.0
for k in sorted(a, key=lambda k: -a[k][0] * a[k][1]):
    if k in b:
        ca, ma, g, fa = a[k]
        cb, mb, _, fb = b[k]
        ta += ca * ma
        tb += cb * mb
        if k[0] != 'wgrad' and ca * ma >= thr:
            print(' '.join(f'{x:>5}' for x in k), f'| {ca:4.1f} {ma:8.4f} {mb:8.4f} {g:7.2f} {fa:7.1f} {fb:7.1f}', '<<' if mb > 1.03 * ma else '')
print('total conv ms/step', round(ta, 3), round(tb, 3))
