#!/bin/bash
# lane-parallel probe (DPP minimum per row instead of per-row scalar extraction): parity, in situ timing, wave-scheduler counters
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests/test_fuzz_gpu.py tests/test_cache_gpu.py tests/test_golden_gpu.py -x -q > gpurun_out/r03/parity_dpp.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r03/parity_dpp.log
(echo "# --- configs[3] shape, avg degree 12 (~289 k rows)"
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 REPS=2 timeout -k 10 900 python tools/k1_insitu.py "" "SINGLE=1" 2>/dev/null | grep -v "^# setup"
echo "# --- configs[3] shape, avg degree 6 (~72 k rows)"
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 REPS=2 timeout -k 10 900 python tools/k1_insitu.py "" "SINGLE=1" 2>/dev/null | grep -v "^# setup"
echo "# --- default workload + all-hit leg"
REPS=2 ALLHIT=1 timeout -k 10 600 python tools/k1_insitu.py "" "SINGLE=1" 2>/dev/null | grep -v "^# setup") > gpurun_out/r03/k1_dpp.txt 2>&1
cat gpurun_out/r03/k1_dpp.txt
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 bash tools/k1_sq_counters.sh papers100m 2>&1 | grep -A30 "^--- K1" | head -32 > gpurun_out/r03/k1_sq_papers100m_dpp.txt; cat gpurun_out/r03/k1_sq_papers100m_dpp.txt
