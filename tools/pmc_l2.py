"""Reduce one rocprofv3 PMC pass (--pmc TCC_HIT_sum TCC_MISS_sum) to the L2 hit rate of every conv / BatchNorm kernel symbol.

    python tools/pmc_l2.py <counter_collection.csv> <out.json> [workload text]

hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)  (MI355X_MICROARCH.md, L2 section), summed over the launches of a symbol.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("udaseg::", "")


def main():
    acc = defaultdict(lambda: defaultdict(float))
    seen, cnt = set(), defaultdict(int)
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        if not k.startswith(("conv", "bn_")):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"]))
            cnt[k] += 1
    out = {}
    for k, v in acc.items():
        h, m = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
        out[k] = {"launches": cnt[k], "tcc_hit_per_launch": round(h / max(cnt[k], 1)), "tcc_miss_per_launch": round(m / max(cnt[k], 1)),
                  "l2_hit_rate": round(h / (h + m), 4) if h + m else None}
    leg = sys.argv[3] if len(sys.argv) > 3 else "workload not stated"
    json.dump({"note": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum, " + leg + ", single stream; requests of 128 B", "kernels": out}, open(sys.argv[2], "w"),
              indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["tcc_hit_per_launch"] + kv[1]["tcc_miss_per_launch"]) * kv[1]["launches"]):
        print(f"{k:60s} n={v['launches']:4d} hit rate {v['l2_hit_rate']}")


if __name__ == "__main__":
    main()
