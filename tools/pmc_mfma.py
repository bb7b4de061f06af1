"""Reduce one rocprofv3 PMC pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA) to the
matrix-pipe utilisation of every conv kernel symbol.

    python tools/pmc_mfma.py <counter_collection.csv> <out.json> [workload text]

SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is reported as the sum over the 8 XCDs
(MI355X_MICROARCH.md, DVFS section), so   mfma_busy = BUSY_CYCLES / (1024 * GUI_ACTIVE / 8)   is the fraction of SIMD cycles
the matrix pipe was executing, clock-independent.  valu_per_mfma = non-MFMA VALU instructions per MFMA instruction
(SQ_INSTS_VALU counts MFMAs too) -- on gfx950 fp32 MFMA shares its ALUs with the VALU, so this ratio bounds mfma_busy.

GRBM_GUI_ACTIVE spans MORE than the kernel for short launches (the counter window opens before the first wave and closes after the
last store has drained): clock_ghz = GUI_ACTIVE / 8 / trace duration came out at up to 3.2 "GHz" on a 2.4 GHz part for launches
under ~60 us in round 4, i.e. mfma_busy of those kernels was understated by up to 25 %.  Every entry therefore also
carries mfma_busy_at_2p4 = BUSY_CYCLES / (1024 * trace duration * 2.4 GHz) -- the fraction of the datasheet-clock pipe cycles of the
kernel's own duration, a LOWER bound that does not depend on the counter window -- and gui_window_suspect = true when clock_ghz
exceeds the part's 2.4 GHz (then mfma_busy's denominator is known to be too long; read mfma_busy_at_2p4 x (2.4 / the real clock)).
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
    cnt = defaultdict(int)
    seen = set()
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        if not k.startswith("conv"):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
            acc[k]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {}
    for k, v in acc.items():
        gui = v.get("GRBM_GUI_ACTIVE", 0.0)
        busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        mf = v.get("SQ_INSTS_MFMA", 0.0)
        out[k] = {"launches": cnt[k], "avg_us": round(v["_ns"] / cnt[k] / 1e3, 1),
                  "mfma_busy": round(busy / (1024 * gui / 8), 4) if gui else None,
                  "valu_per_mfma": round((v.get("SQ_INSTS_VALU", 0.0) - mf) / mf, 2) if mf else None,
                  "clock_ghz": round(gui / 8 / v["_ns"], 3) if v["_ns"] else None,
                  "mfma_busy_at_2p4": round(busy / (1024 * v["_ns"] * 2.4), 4) if v["_ns"] else None,
                  "gui_window_suspect": bool(v["_ns"] and gui / 8 / v["_ns"] > 2.45)}
    leg = sys.argv[3] if len(sys.argv) > 3 else "r18-Unet 8x3x512x512 fp32 train step (BASELINE cfg 2)"
    json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA, " + leg +
                       ", single stream; mfma_busy = fraction of SIMD cycles with the matrix pipe busy (denominator GRBM_GUI_ACTIVE: too "
                       "long for short launches -- gui_window_suspect); mfma_busy_at_2p4 = the same cycles over the kernel's trace "
                       "duration at the datasheet 2.4 GHz (a lower bound, independent of the counter window)",
               "kernels": out}, open(sys.argv[2], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"]):
        print(f"{k:52s} n={v['launches']:4d} {v['avg_us']:8.1f} us  mfma_busy {v['mfma_busy']} (>= {v['mfma_busy_at_2p4']} at 2.4 GHz{', GUI window suspect' if v['gui_window_suspect'] else ''})  valu/mfma {v['valu_per_mfma']}  clk {v['clock_ghz']}")


if __name__ == "__main__":
    main()
