#!/bin/bash
# Engine clock while a kernel runs: GRBM_GUI_ACTIVE (busy cycles of the whole GPU) over the kernel's duration.
# kernel_clock.sh <kernel name substring> -- <python script and arguments>
export TMPDIR=/tmp
PAT=$1; shift; shift
OUT=gpurun_out/clk
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- python3 "$@" > $OUT/log 2>&1
python3 - "$PAT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/clk/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            acc[r["Kernel_Name"][:70]].append((float(r["Counter_Value"]), d))
for k, v in acc.items():
    v = v[len(v) // 4:]                      # (the first launches carry start-up)
    print(k, "launches", len(v), "mean us %.1f" % (sum(d for _, d in v) / len(v) / 1e3),
          "engine clock GHz (counter summed over the 8 XCDs / 8 / duration): mean %.3f min %.3f max %.3f" % (
              sum(c for c, _ in v) / sum(d for _, d in v) / 8, min(c / d for c, d in v) / 8, max(c / d for c, d in v) / 8))
PY
