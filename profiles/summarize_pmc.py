#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (profiles/collect_pmc.sh): per-kernel mean of every counter
and mean kernel duration.  FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950
FETCH_SIZE under-counts wide coalesced streams by 2x (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, os, sys
from collections import defaultdict

def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return n.split("(")[0][:28]

def main(root):
    res = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for d in sorted(glob.glob(os.path.join(root, "*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                res[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")):
            for r in csv.DictReader(open(f)):
                dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k in sorted(res, key=lambda k: -sum(dur.get(k, [0])) / max(len(dur.get(k, [1])), 1)):
        if not k.startswith("k_"):
            continue
        row = {c: sum(v) / len(v) for c, v in res[k].items()}
        row["us(profiled)"] = sum(dur[k]) / len(dur[k]) if dur.get(k) else 0
        out[k] = row
        print(k, {c: (round(v, 1) if v < 1e4 else float("%.4g" % v)) for c, v in sorted(row.items())})
    return out

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
