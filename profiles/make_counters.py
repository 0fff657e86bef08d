#!/usr/bin/env python3
"""Per-stage counter summary of one workload from the passes of collect_pmc.sh, merged into profiles/counters.json
(what bench.py reports as roofline.traffic / counter_frac / valu).

usage: make_counters.py <pmc dir> <tag> <workload>      e.g.  make_counters.py gpurun_out/pmc_C5 r02_v3 C5

Writes profiles/<tag>_<workload>_pmc_traffic.json (HBM bytes per launch per kernel = factor * FETCH_SIZE + WRITE_SIZE with
the factor of the kernel's access kind as calibrated on this box -- profiles/fetch_factors.py: 2 for wide coalesced
streams, which FETCH_SIZE counts at half their bytes (MI355X_MICROARCH.md HBM section), 1 for the record gathers of the
blend kernels, whose 64-byte lines it counts exactly; both counters are reported in KiB), profiles/<tag>_<workload>_sq_table.json (SQ passes: resident waves per SIMD, VALU issue share,
wait / stall shares, VALU wave-instructions per launch) and the entry counters.json[workload].  A stage that runs
several kernels per step (tile_sort = head + tail launch, render_fwd = first pass + resume launch) sums them.
"""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fetch_factors

HERE = os.path.dirname(os.path.abspath(__file__))
CLOCK_GHZ, SIMDS = 2.4, 1024
STAGE_OF = {"k_preprocess_fwd": "preprocess_fwd", "k_appearance": "preprocess_fwd",
            "k_tile_count": "tile_count", "k_tile_scatter": "tile_scatter",
            "k_tile_sort_small": "tile_sort", "k_tile_sort_big": "tile_sort", "k_tile_front": "tile_sort", "k_tile_tail": "tile_sort",
            "k_render_fwd": "render_fwd", "k_render_fwd_seg": "render_fwd", "k_render_bwd": "render_bwd", "k_tile_order": "render_bwd",
            "k_preprocess_bwd": "preprocess_bwd", "k_preprocess_bwd_common": "preprocess_bwd", "k_offset_reduce": "preprocess_bwd",
            # gradient tensors kept between backwards: rows of blended Gaussians only (the training call's kernel)
            "k_preprocess_bwd_rows": "preprocess_bwd", "k_grads_rezero": "preprocess_bwd",
            # tile-pull binning (k_pull.hip): count pass, scatter pass, per-tile pull + sort, lists completed on demand
            "k_super_bin<0>": "tile_count", "k_super_bin<1>": "tile_scatter",
            # (<2>: the scatter by the camera's list schedule, no count pass in front)
            "k_super_bin<2>": "tile_scatter", "k_tile_pull": "tile_sort", "k_tail_build": "tile_sort"}
ALL_STAGES = ("preprocess_fwd", "tile_count", "tile_scatter", "tile_sort", "render_fwd", "render_bwd", "preprocess_bwd")


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if n.startswith("k_super_bin<"):                                    # (the two passes of k_super_bin are two stages;
        return "k_super_bin<%s>" % n[len("k_super_bin<")]               #  k_super_bin<1, true> = the scatter pass through LDS)
    return n.split("<")[0]


def source_sha():
    """sha1 over the kernel sources the counters were taken from (bench.py compares it with the sources it runs)"""
    import hashlib
    h = hashlib.sha1()
    root = os.path.join(os.path.dirname(HERE), "gftorf_amd", "csrc")
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


def read(root, sub):
    """kernel -> counter -> [values per dispatch], kernel -> [durations us], kernel -> dispatches"""
    val, dur = defaultdict(lambda: defaultdict(list)), defaultdict(list)
    for f in glob.glob(os.path.join(root, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            val[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(root, sub, "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return val, dur


def main():
    root, tag, workload = sys.argv[1], sys.argv[2], sys.argv[3]
    mean = lambda v: sum(v) / len(v) if v else 0.0
    fetch, _ = read(root, "fetch")
    write, _ = read(root, "write")
    traffic = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_") or "deform" in k:
            continue
        f, w = mean(fetch[k]["FETCH_SIZE"]) * 1024, mean(write[k]["WRITE_SIZE"]) * 1024
        hb, kind, fac = fetch_factors.hbm_bytes(k, f, w)
        traffic[k] = {"FETCH_SIZE_bytes_raw": f, "WRITE_SIZE_bytes": w, "hbm_bytes_corrected": hb, "access_kind": kind,
                      "fetch_factor": fac, "hbm_bytes_if_all_streams": 2 * f + w,
                      "launches_seen": len(fetch[k]["FETCH_SIZE"]) or len(write[k]["WRITE_SIZE"])}
    t_name = "%s_%s_pmc_traffic.json" % (tag, workload)
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (collect_pmc.sh), mean per launch over "
                       "bench.py --workload %s --steps 3 --warmup 1; hbm_bytes = fetch_factor*FETCH_SIZE + WRITE_SIZE, the factor by "
                       "the kernel's access kind (profiles/fetch_factors.py, calibrated: profiles/r06_fetch_calibration.json)" % workload,
               "per_kernel": traffic}, open(os.path.join(HERE, t_name), "w"), indent=1)

    v1, d1 = read(root, "sq1")
    v2, d2 = read(root, "sq2")
    sq = {}
    for k in sorted(set(v1) | set(v2)):
        if not k.startswith("k_") or "deform" in k:
            continue
        m = {c: mean(v) for c, v in list(v1[k].items()) + list(v2[k].items())}
        us = mean(d1.get(k, []) + d2.get(k, []))
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        res = wc * 4 / (us * 1e-6 * CLOCK_GHZ * 1e9) / SIMDS if us else None
        vif = m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc if wc else None
        sq[k] = {"us_profiled": us, "waves": m.get("SQ_WAVES"), "mean_resident_waves_per_simd": res, "valu_issue_frac": vif,
                 "valu_issue_slot_frac": (res * vif) if (res is not None and vif is not None) else None,
                 "lds_issue_frac": m.get("SQ_ACTIVE_INST_LDS", 0.0) / wc if wc else None,
                 "wait_frac": m.get("SQ_WAIT_ANY", 0.0) / wc if wc else None,
                 "stall_frac": m.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else None,
                 "valu_insts": m.get("SQ_INSTS_VALU"), "salu_insts": m.get("SQ_INSTS_SALU"), "lds_insts": m.get("SQ_INSTS_LDS"),
                 "vmem_rd_insts": m.get("SQ_INSTS_VMEM_RD"), "vmem_wr_insts": m.get("SQ_INSTS_VMEM_WR")}
    s_name = "%s_%s_sq_table.json" % (tag, workload)
    json.dump({"note": "SQ passes of collect_pmc.sh (sq1, sq2), mean per launch.  mean_resident_waves_per_simd = SQ_WAVE_CYCLES*4 / "
                       "(duration * 2.4 GHz) / 1024 SIMDs (of 8 slots); valu_issue_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES; "
                       "valu_issue_slot_frac = their product = share of every SIMD's VALU issue slots that are taken",
               "per_kernel": sq}, open(os.path.join(HERE, s_name), "w"), indent=1)

    # per stage: the kernels of a stage summed, weighted by their launches per step
    stages = defaultdict(lambda: {"hbm_bytes": 0.0, "valu_insts": 0.0, "us_profiled": 0.0, "kernels": [], "fetch_factors": {}})
    # launches per forward call: count dispatches relative to k_preprocess_fwd in the same pass
    ref_n = max(len(d1.get("k_preprocess_fwd", [])), 1)
    for k, st in STAGE_OF.items():
        if k not in traffic and k not in sq:
            continue
        per_step = (len(d1.get(k, [])) / ref_n) if d1.get(k) else 1.0
        s = stages[st]
        s["kernels"].append(k)
        s["hbm_bytes"] += traffic.get(k, {}).get("hbm_bytes_corrected", 0.0) * per_step
        if k in traffic:
            s["fetch_factors"][k] = "%s x%g" % (traffic[k]["access_kind"], traffic[k]["fetch_factor"])
        if k in sq:
            s["valu_insts"] += (sq[k]["valu_insts"] or 0.0) * per_step
            s["us_profiled"] += sq[k]["us_profiled"] * per_step
    for st in ALL_STAGES:
        stages[st]                       # a stage none of whose kernels ran keeps a zero entry
    for st, s in stages.items():
        if not s["kernels"]:
            s["dominant_kernel"] = None
            continue
        main_k = max(s["kernels"], key=lambda k: sq.get(k, {}).get("us_profiled", 0.0))
        s["dominant_kernel"] = main_k
        for f in ("mean_resident_waves_per_simd", "valu_issue_frac", "valu_issue_slot_frac", "wait_frac", "stall_frac"):
            s[f] = sq.get(main_k, {}).get(f)
    cpath = os.path.join(HERE, "counters.json")
    doc = json.load(open(cpath)) if os.path.exists(cpath) else {}
    doc[workload] = {"source": ["profiles/" + t_name, "profiles/" + s_name], "kernel_source_sha": source_sha(), "stages": stages}
    json.dump(doc, open(cpath, "w"), indent=1)
    for st, s in stages.items():
        print("%-15s hbm %8.1f MB  valu insts %11.0f  %7.1f us  waves/SIMD %s  VALU slots %s" % (
            st, s["hbm_bytes"] / 1e6, s["valu_insts"], s["us_profiled"], s.get("mean_resident_waves_per_simd"), s.get("valu_issue_slot_frac")))


if __name__ == "__main__":
    main()
