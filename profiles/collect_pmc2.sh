#!/bin/bash
OUT=gpurun_out/pmc2; export TMPDIR=/tmp; mkdir -p $OUT
run() { local name=$1; shift; timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/$name.log" 2>&1; echo "pass $name rc=$?"; }
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run fetch FETCH_SIZE
run write WRITE_SIZE
