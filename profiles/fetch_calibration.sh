#!/bin/bash
# FETCH_SIZE per access kind on this box (profiles/experiments/fetch_calibration.hip): run on the GPU box from the repo root.
export TMPDIR=/tmp
set -u
OUT=gpurun_out/fetch_calib
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib profiles/experiments/fetch_calibration.hip || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- /tmp/fetch_calib > $OUT/run.log 2>&1
echo "rc=$?"; tail -2 $OUT/run.log
python3 - <<'PY'
import csv, glob, json
true = {"k_stream16": 64, "k_stream4": 64, "k_gather<2>": 32, "k_gather32x2": 64, "k_gather<4>": 64, "k_gather<12>": 192}
n = 1 << 24
acc = {}
for f in glob.glob("gpurun_out/fetch_calib/fetch/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            acc.setdefault(name, []).append(float(r["Counter_Value"]) * 1024)
out = {}
for k, v in sorted(acc.items()):
    if k in true:
        rep = sum(v) / len(v)
        out[k] = {"bytes_read": true[k] * n, "FETCH_SIZE_bytes": rep, "bytes_per_reported_byte": true[k] * n / rep if rep else None}
        print("%-14s read %6.0f MB  FETCH_SIZE %7.1f MB  -> x%.3f" % (k, true[k] * n / 1e6, rep / 1e6, true[k] * n / rep))
json.dump(out, open("gpurun_out/r06_fetch_calibration.json", "w"), indent=1)
PY
rm -rf $OUT/fetch
