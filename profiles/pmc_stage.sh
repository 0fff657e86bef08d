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
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "render" not in k:
            continue
        k = "k_render_fwd_seg" if "fwd_seg" in k else ("k_render_bwd" if "bwd" in k else "k_render_fwd")
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_WAVES":
            rows[k]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, c in rows.items():
    n = len(c["SQ_WAVES"])
    busy = [i for i in range(n) if c["SQ_INSTS_VALU"][i] > 0]        # (idle resume launches left out)
    if not busy:
        continue
    m = {a: sum(b[i] for i in busy) / len(busy) for a, b in c.items()}
    dur = m["dur_us"]
    # SQ_WAVE_CYCLES / SQ_BUSY_CYCLES count in units of 4 clocks on this part (a resident serial wave of 100 us reads ~60 k)
    print(k, "launches", len(busy), "dur_us %.1f" % dur, "waves %d" % m["SQ_WAVES"], "VALU_wave_instr %.2fM" % (m["SQ_INSTS_VALU"] / 1e6),
          "SALU %.2fM" % (m["SQ_INSTS_SALU"] / 1e6),
          "valu_issue_share %.2f" % (m["SQ_INSTS_VALU"] * 4 / (1024 * dur * 2400)),
          "mean_wave_life_us %.1f" % (m["SQ_WAVE_CYCLES"] * 4 / m["SQ_WAVES"] / 2400),
          "mean_resident_waves_per_simd %.2f" % (m["SQ_WAVE_CYCLES"] * 4 / 2400 / dur / 1024))
PY
