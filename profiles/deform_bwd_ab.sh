#!/bin/bash
# A/B of the deformation network's backward walk: two fp16 planes (default) against three bf16 planes
# (GFT_DEFORM_BWD_FP16=0), kernel times from rocprofv3.  Usage (GPU box): bash profiles/deform_bwd_ab.sh [points...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for n in "${@:-300000}"; do
  for m in 1 0; do
    export GFT_DEFORM_BWD_FP16=$m
    rm -rf /tmp/dab
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dab -- python3 $R/profiles/deform_ab.py $n > /tmp/dab.log 2>&1 || { tail -5 /tmp/dab.log; exit 1; }
    echo "== points=$n GFT_DEFORM_BWD_FP16=$m"; grep '^{' /tmp/dab.log
    f=$(find /tmp/dab -name '*kernel_stats.csv' | head -1)
    python3 - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'deform' in r['Name']:
        print("   %-60s calls %4s avg %9.1f us" % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
P
  done
done
