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
import csv, glob, collections, json
res = {}
for f in glob.glob("gpurun_out/dprof/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "deform" in r["Name"]:
            res.setdefault(r["Name"].split("(")[0].split("::")[-1].replace("void ", ""), {})["avg_us"] = float(r["AverageNs"]) / 1e3
for sub in ("sq1", "fetch", "write"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for f in glob.glob("gpurun_out/dprof/%s/**/*counter_collection.csv" % sub, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if "deform" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", "")
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if (r["Dispatch_Id"]) not in seen:
                    seen.add(r["Dispatch_Id"]); calls[k] += 1
    for k, d in agg.items():
        for c, v in d.items():
            res.setdefault(k, {})[c] = v / max(calls[k], 1)
json.dump(res, open("gpurun_out/dprof/deform_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
