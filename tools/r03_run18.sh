#!/bin/bash
# does K1 start on a chip that has clocked down during the PCIe-bound fill?  compute-bound kernels between the fill of one step and the K1 of the next
set -o pipefail
mkdir -p gpurun_out/r03
(echo "# --- configs[3] shape, avg degree 6 (~72 k rows): the box-speed indicator (22 us on some boxes, 35-37 us on others)"
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 REPS=2 timeout -k 10 900 python tools/k1_insitu.py "" "BUSY=1" "BUSY=4" "BUSY=16" 2>/dev/null | grep -v "^# setup"
echo "# --- configs[3] shape, avg degree 12 (~289 k rows)"
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 REPS=2 timeout -k 10 900 python tools/k1_insitu.py "" "BUSY=4" "BUSY=16" 2>/dev/null | grep -v "^# setup"
echo "# --- default workload (4-KiB lines, 28.5 k rows)"
REPS=2 timeout -k 10 600 python tools/k1_insitu.py "" "BUSY=4" "BUSY=16" 2>/dev/null | grep -v "^# setup") > gpurun_out/r03/k1_busy.txt 2>&1
cat gpurun_out/r03/k1_busy.txt
rocm-smi --showclocks --showperflevel --showpower 2>/dev/null | head -30
