#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/dprof
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 profiles/deform_workload.py > $OUT/stats.log 2>&1; echo stats rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq1 -- python3 profiles/deform_workload.py > $OUT/sq1.log 2>&1; echo sq1 rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 profiles/deform_workload.py > $OUT/fetch.log 2>&1; echo fetch rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 profiles/deform_workload.py > $OUT/write.log 2>&1; echo write rc=$?
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
python3 - <<'PY'
import csv, glob, collections, json, re
def kname(n):
    m = re.search(r"(k_deform_[a-z_]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or "")) if m else None
res = {}
for f in glob.glob("gpurun_out/dprof/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = kname(r["Name"])
        if k: res.setdefault(k, {}).update(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3)
for sub in ("sq1", "fetch", "write"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in glob.glob("gpurun_out/dprof/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k, d in agg.items():
        for c, v in d.items():
            res.setdefault(k, {})[c] = v / max(len(disp[k]), 1)
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:      # KiB per launch; gfx950 correction as in make_traffic.py
        d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
json.dump(res, open("gpurun_out/dprof/deform_counters.json", "w"), indent=1)
for k, d in res.items(): print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in d.items()})
PY
