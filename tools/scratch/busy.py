"""GPU busy fraction over the last steps of a rocprofv3 kernel trace: union of kernel intervals / span."""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
adam = [i for i, r in enumerate(iv) if "adam_flat" in r[2]]
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 3
per = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # adam launches per step
lo, hi = adam[-1 - nst * per] + 1, adam[-1] + 1
seg = iv[lo:hi]
t0, t1 = seg[0][0], max(e for _, e, _ in seg)
busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
gaps = []
for s, e, n in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{nst} steps: span {(t1 - t0) / 1e6 / nst:.3f} ms/step, busy {busy / 1e6 / nst:.3f} ms/step = {100 * busy / (t1 - t0):.1f} %, {len(seg) // nst} launches/step")
gaps.sort(reverse=True)
tot = sum(g for g, _ in gaps)
print(f"idle {tot / 1e6 / nst:.3f} ms/step in {len(gaps) // nst} gaps/step; gaps > 5 us: {sum(1 for g, _ in gaps if g > 5000) // nst}/step = {sum(g for g, _ in gaps if g > 5000) / 1e6 / nst:.3f} ms")
for g, n in gaps[:12]:
    print(f"   {g / 1e3:8.1f} us before {n[:90]}")
