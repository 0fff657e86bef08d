#!/bin/bash
# Kernel statistics of the composed C4 step and of the deformation network alone: collect_c4.sh <tag>
TAG=${1:-r04}
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_${TAG}_C4 -- python3 bench.py --workload C4 --no-cpu-baseline --steps 60 --warmup 10 > gpurun_out/${TAG}_C4_stats_bench.json 2> gpurun_out/${TAG}_C4_stats.err
f=$(ls gpurun_out/stats_${TAG}_C4/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_C4_kernel_stats.csv && head -12 "$f" | cut -c1-150
rm -rf gpurun_out/stats_${TAG}_C4
GFT_DEFORM_LAZY_SAVE=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_${TAG}_deform -- python3 profiles/deform_workload.py > gpurun_out/${TAG}_deform_stats.log 2>&1
f=$(ls gpurun_out/stats_${TAG}_deform/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_deform_kernel_stats.csv && head -10 "$f" | cut -c1-150
rm -rf gpurun_out/stats_${TAG}_deform
bash profiles/pmc_deform_fwd.sh gpurun_out/pmc_${TAG}_dfwd > gpurun_out/${TAG}_deform_fwd_counters.txt 2>&1
cp gpurun_out/pmc_${TAG}_dfwd/deform_fwd_counters.json gpurun_out/${TAG}_deform_fwd_counters.json
rm -rf gpurun_out/pmc_${TAG}_dfwd
tail -50 gpurun_out/${TAG}_deform_fwd_counters.txt | head -60
