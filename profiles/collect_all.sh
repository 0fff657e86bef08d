#!/bin/bash
# Counter passes + kernel statistics for the workloads bench.py reports counters for.
# usage: collect_all.sh <tag> [workloads...]      (run on the GPU box from the repo root)
TAG=${1:-r03_v1}
shift
WLS=${@:-metric C2 C5}
export TMPDIR=/tmp
for WL in $WLS; do
  echo "== $WL"
  bash profiles/collect_pmc.sh gpurun_out/pmc_${TAG}_$WL $WL
  python3 profiles/make_counters.py gpurun_out/pmc_${TAG}_$WL $TAG $WL > gpurun_out/${TAG}_${WL}_counters.txt 2>&1
  tail -12 gpurun_out/${TAG}_${WL}_counters.txt
  rm -rf gpurun_out/pmc_${TAG}_$WL
  bash profiles/kstats.sh $TAG $WL > gpurun_out/${TAG}_${WL}_kstats.log 2>&1
  cp gpurun_out/${TAG}_${WL}_kernel_stats.csv profiles/${TAG}_${WL}_kernel_stats.csv 2>/dev/null
  cp gpurun_out/${TAG}_${WL}_views_kernel_stats.csv profiles/${TAG}_${WL}_views_kernel_stats.csv 2>/dev/null
done
cp profiles/counters.json gpurun_out/${TAG}_counters.json
cp profiles/${TAG}_* gpurun_out/ 2>/dev/null
