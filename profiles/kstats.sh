#!/bin/bash
# rocprofv3 kernel statistics of the repeated-frame step and of the varying-view step.
# usage: kstats.sh <tag> [workload]      (run on the GPU box from the repo root; writes gpurun_out/<tag>_*)
TAG=${1:-r03}
WL=${2:-metric}
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_${TAG}_${WL} -- python3 bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_${WL}_stats_bench.json 2> gpurun_out/${TAG}_${WL}_stats.err
f=$(ls gpurun_out/stats_${TAG}_${WL}/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_${WL}_kernel_stats.csv && head -16 "$f"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_${TAG}_${WL}_views -- python3 profiles/views_workload.py $WL 60 > gpurun_out/${TAG}_${WL}_views_bench.json 2> gpurun_out/${TAG}_${WL}_views.err
f=$(ls gpurun_out/stats_${TAG}_${WL}_views/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_${WL}_views_kernel_stats.csv && head -16 "$f"
rm -rf gpurun_out/stats_${TAG}_${WL} gpurun_out/stats_${TAG}_${WL}_views
