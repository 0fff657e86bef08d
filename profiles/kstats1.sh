#!/bin/bash
# rocprofv3 kernel statistics of the repeated-frame step only.  usage: kstats1.sh <tag> [workload]
TAG=${1:-r03}
WL=${2:-metric}
export TMPDIR=/tmp
timeout -k 10 ${KSTATS_TIMEOUT:-120} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_${TAG}_${WL} -- python3 bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/${TAG}_${WL}_stats_bench.json 2> gpurun_out/${TAG}_${WL}_stats.err || exit 1
f=$(ls gpurun_out/stats_${TAG}_${WL}/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_${WL}_kernel_stats.csv && head -12 "$f" | cut -c1-120 | grep -v "at::native\|rocclr"
rm -rf gpurun_out/stats_${TAG}_${WL}
