#!/bin/bash
# Collect PMC counters for every kernel of one bench run (separate passes; --pmc is never
# combined with tracing options other than --kernel-trace).  Usage: collect_pmc.sh <outdir> [workload]
set -u
OUT=${1:-gpurun_out/pmc}
WL=${2:-metric}
export TMPDIR=/tmp
mkdir -p "$OUT"
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
      python3 bench.py --workload "$WL" --steps 3 --warmup 1 --spin-up 0.05 --no-cpu-baseline --no-extras > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum
