#!/bin/bash
# rocprofv3 kernel statistics + SQ counters of the rasterizer on the C3-shaped frame (100 k Gaussians, 320x240, opacity 0.1):
# usage: kstats_c3.sh <tag>      (run on the GPU box from the repo root; GFT_FWD_SEG passes through)
TAG=${1:-r04}
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_${TAG}_c3 -- python3 profiles/stage_bench.py c3 100 > gpurun_out/${TAG}_c3_stats_bench.json 2> gpurun_out/${TAG}_c3_stats.err
f=$(ls gpurun_out/stats_${TAG}_c3/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_c3frame_kernel_stats.csv && head -14 "$f" | cut -c1-140
rm -rf gpurun_out/stats_${TAG}_c3
bash profiles/pmc_stage.sh gpurun_out/pmc_${TAG}_c3 c3 > gpurun_out/${TAG}_c3frame_sq.txt 2>&1
tail -4 gpurun_out/${TAG}_c3frame_sq.txt
rm -rf gpurun_out/pmc_${TAG}_c3
