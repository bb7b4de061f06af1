"""Reduce two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes)
to HBM bytes per launch for every kernel symbol.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [workload text]

Corrections (guide, section HBM): counter values are KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
(16 B/lane) coalesced read stream, so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores and float
atomics.  traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes.
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


def collect(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


def main():
    f_tot, f_cnt = collect(sys.argv[1], "FETCH_SIZE")
    w_tot, w_cnt = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f_tot) | set(w_tot)):
        if not k.startswith(("conv", "bn_", "ce_", "adam", "upcat", "maxpool", "channel", "pack", "small", "nchw", "gap", "bce")):
            continue
        nf, nw = max(f_cnt.get(k, 0), 1), max(w_cnt.get(k, 0), 1)
        fetch_kib, write_kib = f_tot.get(k, 0.0) / nf, w_tot.get(k, 0.0) / nw
        out[k] = {"launches": f_cnt.get(k, 0), "fetch_kib_raw": round(fetch_kib, 1), "write_kib": round(write_kib, 1),
                  "hbm_bytes_per_launch": int((2 * fetch_kib + write_kib) * 1024)}
    what = sys.argv[4] if len(sys.argv) > 4 else "r18-Unet 8x3x512x512 fp32 train step"
    json.dump({"note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), {what}, "
                       "per-launch averages; read side x2 (gfx950 FETCH_SIZE under-count), KiB -> bytes",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k:44s} n={v['launches']:4d} fetch {v['fetch_kib_raw']:10.1f} KiB write {v['write_kib']:10.1f} KiB  -> "
              f"{v['hbm_bytes_per_launch'] / 1e6:8.2f} MB/launch")


if __name__ == "__main__":
    main()
