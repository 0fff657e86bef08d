#!/bin/bash
# HBM bytes per iteration of the colour + ToF camera pair: two calls + autograd's sum against GaussianRasterizerPair.
# Separate --pmc passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only; every kernel of the process is summed (the
# two-call variant's extra bytes are torch's elementwise adds of the dense gradient sets and the second set of zero rows).
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pairpmc}
N=10
mkdir -p $OUT
for W in two pair; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${W}_$C -- python3 profiles/pair_workload.py $W $N > $OUT/${W}_$C.log 2>&1
    echo "$W $C rc=$?"
  done
done
python3 - <<PY
import csv, glob, json, collections
N, WARM = $N, 5
res = {}
for w in ("two", "pair"):
    tot = {}
    per_kernel = collections.defaultdict(float)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        s = 0.0
        for f in glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (w, c), recursive=True):
            for r in csv.DictReader(open(f)):
                v = float(r["Counter_Value"]) * 1024 * (2 if c == "FETCH_SIZE" else 1)     # KiB; gfx950: FETCH_SIZE counts half
                s += v
                per_kernel[r["Kernel_Name"].split("(")[0][-60:]] += v
        tot[c] = s
    iters = N + WARM
    res[w] = {"hbm_bytes_per_iteration": (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / iters,
              "fetch_bytes_per_iteration": tot["FETCH_SIZE"] / iters, "write_bytes_per_iteration": tot["WRITE_SIZE"] / iters,
              "largest_kernels_bytes_per_iteration": {k: v / iters for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1])[:12]}}
res["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), all kernels of profiles/pair_workload.py summed over %d "
               "iterations (incl. 5 warm-up, scene upload excluded by being outside kernels); HBM bytes = 2*FETCH_SIZE + WRITE_SIZE (KiB units)" % (N + WARM))
res["saved_bytes_per_iteration"] = res["two"]["hbm_bytes_per_iteration"] - res["pair"]["hbm_bytes_per_iteration"]
json.dump(res, open("$OUT/pair_pmc.json", "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "largest_kernels_bytes_per_iteration"}) for k, v in res.items()}, indent=1))
PY
