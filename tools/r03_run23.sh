#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r03/gputest_c.log 2>&1; echo "gputests rc=$?"; tail -14 gpurun_out/r03/gputest_c.log
(echo "# --- configs[3] shape, avg degree 12 (~289 k rows)"
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 REPS=2 timeout -k 10 900 python tools/k1_insitu.py "" "PASSES=8" "PASSES=2" "WAVES=4" 2>/dev/null | grep -v "^# setup"
echo "# --- configs[3] shape, avg degree 6 (~72 k rows)"
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 REPS=2 timeout -k 10 900 python tools/k1_insitu.py "" "PASSES=8" "PASSES=2" 2>/dev/null | grep -v "^# setup"
echo "# --- default workload + all-hit leg"
REPS=2 ALLHIT=1 timeout -k 10 600 python tools/k1_insitu.py "" "TAG64=1" 2>/dev/null | grep -v "^# setup") > gpurun_out/r03/k1_dpp2.txt 2>&1
cat gpurun_out/r03/k1_dpp2.txt
