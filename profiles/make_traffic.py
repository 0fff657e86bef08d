#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of collect_pmc.sh into per-kernel HBM bytes per launch.

usage: make_traffic.py <pmc dir> <out json> [--update-traffic-json]

rocprofv3 reports both counters in KiB.  On gfx950 FETCH_SIZE counts wide coalesced reads at
half their size (MI355X_MICROARCH.md, HBM / rocprofv3 section) and the 64-byte lines of scattered record
reads exactly (calibrated: profiles/fetch_factors.py), hence hbm = factor(kernel) * FETCH + WRITE.
"""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fetch_factors

STAGE_OF = {"k_preprocess_fwd": "preprocess_fwd", "k_tile_count": "tile_count", "k_tile_scatter": "tile_scatter",
            "k_tile_sort_small": "tile_sort", "k_tile_front": "tile_sort", "k_render_fwd": "render_fwd", "k_render_bwd": "render_bwd",
            "k_preprocess_bwd": "preprocess_bwd", "k_preprocess_bwd_common": "preprocess_bwd"}


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].split("<")[0]


def mean_counter(root, name, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(root, name, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    root, out = sys.argv[1], sys.argv[2]
    fetch = mean_counter(root, "fetch", "FETCH_SIZE")
    write = mean_counter(root, "write", "WRITE_SIZE")
    per = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, w = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        hb, kind, fac = fetch_factors.hbm_bytes(k, f, w)
        per[k] = {"FETCH_SIZE_bytes_raw": f, "WRITE_SIZE_bytes": w, "hbm_bytes_corrected": hb, "access_kind": kind, "fetch_factor": fac}
    doc = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, mean per launch over bench.py "
                   "--steps 3 --warmup 1 (metric workload); hbm_bytes = fetch_factor*FETCH_SIZE + WRITE_SIZE (profiles/fetch_factors.py)",
           "per_kernel": per}
    json.dump(doc, open(out, "w"), indent=1)
    if "--update-traffic-json" in sys.argv:
        here = os.path.dirname(os.path.abspath(__file__))
        t = {"metric": {STAGE_OF[k]: int(v["hbm_bytes_corrected"]) for k, v in per.items() if k in STAGE_OF}}
        json.dump(t, open(os.path.join(here, "traffic.json"), "w"), indent=1)
    for k, v in per.items():
        print("%-20s fetch %8.1f MB  write %8.1f MB  hbm %8.1f MB" % (k, v["FETCH_SIZE_bytes_raw"] / 1e6,
                                                                   v["WRITE_SIZE_bytes"] / 1e6, v["hbm_bytes_corrected"] / 1e6))


if __name__ == "__main__":
    main()
