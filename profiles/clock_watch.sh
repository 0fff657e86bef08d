#!/bin/bash
# Samples rocm-smi's engine clock and power while a workload runs: clock_watch.sh <python script and arguments>
( for i in $(seq 1 12); do sleep 1.0; rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|power" | tr '\n' ' '; echo; done ) &
W=$!
python3 "$@" > /dev/null 2>&1
wait $W
