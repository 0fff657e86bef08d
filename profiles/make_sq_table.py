#!/usr/bin/env python3
"""Per-kernel occupancy / issue statistics from the SQ passes of collect_pmc.sh.

usage: make_sq_table.py <pmc dir> <out json>

Counters are summed over the chip by rocprofv3.  SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count
quad-cycles (4 shader cycles), SQ_BUSY_CYCLES counts per shader engine (MI355X_MICROARCH.md, PMC section).
Reported per launch:
  waves                     SQ_WAVES
  mean_resident_waves_per_simd = SQ_WAVE_CYCLES * 4 / (duration * 2.4 GHz) / 1024 SIMDs   (of 8 slots)
  valu_issue_frac           SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES  (share of a wave's life spent issuing VALU)
  wait_frac / stall_frac    SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt, barriers), SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
"""
import csv, glob, json, os, sys
from collections import defaultdict

CLOCK_GHZ, SIMDS = 2.4, 1024


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]


def main():
    root, out = sys.argv[1], sys.argv[2]
    val, dur = defaultdict(lambda: defaultdict(list)), defaultdict(list)
    for sub in ("sq1", "sq2"):
        for f in glob.glob(os.path.join(root, sub, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                val[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(os.path.join(root, sub, "*", "*_kernel_trace.csv")):
            for r in csv.DictReader(open(f)):
                dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    table = {}
    for k in sorted(val):
        if not k.startswith("k_") or "deform" in k or not dur.get(k):
            continue
        m = {c: sum(v) / len(v) for c, v in val[k].items()}
        us = sum(dur[k]) / len(dur[k])
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        row = {"us_profiled": us, "waves": m.get("SQ_WAVES"),
               "mean_resident_waves_per_simd": wc * 4 / (us * 1e-6 * CLOCK_GHZ * 1e9) / SIMDS if us else None,
               "valu_issue_frac": m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc if wc else None,
               "lds_issue_frac": m.get("SQ_ACTIVE_INST_LDS", 0.0) / wc if wc else None,
               "wait_frac": m.get("SQ_WAIT_ANY", 0.0) / wc if wc else None,
               "stall_frac": m.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else None,
               "valu_insts": m.get("SQ_INSTS_VALU"), "lds_insts": m.get("SQ_INSTS_LDS"),
               "lds_bank_conflict_cycles": m.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": m.get("SQ_LDS_IDX_ACTIVE")}
        table[k] = row
        print("%-22s %7.1f us  waves/SIMD %.2f  VALU %.2f  LDS %.2f  wait %.2f  stall %.2f" % (
            k, us, row["mean_resident_waves_per_simd"] or 0, row["valu_issue_frac"] or 0, row["lds_issue_frac"] or 0,
            row["wait_frac"] or 0, row["stall_frac"] or 0))
    json.dump({"note": __doc__, "per_kernel": table}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
