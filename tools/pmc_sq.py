"""Reduce a rocprofv3 --pmc pass of SQ counters to per-kernel-symbol ratios (wave time split, LDS conflict share, MFMA busy).

    python tools/pmc_sq.py <counter_collection.csv> [symbol-prefix]

SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES (quad-cycles); SQ_VALU_MFMA_BUSY_CYCLES is in cycles summed
over 1024 SIMDs, GRBM_GUI_ACTIVE summed over 8 XCDs (MI355X_MICROARCH.md).
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("udaseg::", "")


acc = defaultdict(lambda: defaultdict(float))
pref = sys.argv[2] if len(sys.argv) > 2 else "conv"
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"])
    if k.startswith(pref):
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    gui = max(v.get("GRBM_GUI_ACTIVE", 0.0), 1.0)
    line = [f"{k[:58]:58s}"]
    for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU",
                 "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC"):
        if name in v:
            line.append(f"{name[3:]}={v[name] / wc:.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        line.append(f"mfma_busy={v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * gui / 8):.3f}")
    if "SQ_LDS_IDX_ACTIVE" in v:
        line.append(f"lds_conflict_share={v.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(v['SQ_LDS_IDX_ACTIVE'], 1.0):.3f}")
        line.append(f"lds_active/gui={v['SQ_LDS_IDX_ACTIVE'] / (256 * gui / 8):.3f}")
    print("  ".join(line))
