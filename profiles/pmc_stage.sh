#!/bin/bash
# SQ counters of the render kernels on one stage_bench frame: pmc_stage.sh <outdir> <frame> (environment switches pass through)
set -u
OUT=${1:-gpurun_out/pmc_stage}
FR=${2:-c3}
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d "$OUT/sq" -- python3 profiles/stage_bench.py "$FR" 5 > "$OUT/sq.log" 2>&1
echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(out + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "render" not in k:
            continue
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            n[k] += 1
for k, c in rows.items():
    d = max(n[k], 1)
    print(k, "launches", n[k], {a: round(b / d) for a, b in c.items()})
PY
