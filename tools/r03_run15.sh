#!/bin/bash
# two-kernel K1 (probe, then dense copy of the hits) against the fused product kernel, in situ, output rotating over three buffers
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests/test_fuzz_gpu.py -x -q -k "other_launch_shapes and (12 or 13)" > gpurun_out/r03/fuzz_split.log 2>&1; echo "fuzz split rc=$?"; tail -3 gpurun_out/r03/fuzz_split.log
(echo "# --- configs[3] shape, avg degree 12 (~289 k rows)"
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 REPS=2 timeout -k 10 900 python tools/k1_insitu.py "" "SPLIT=1" "SPLIT=2" "SPLIT=4" 2>/dev/null | grep -v "^# setup"
echo "# --- configs[3] shape, avg degree 6 (~72 k rows)"
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 REPS=2 timeout -k 10 900 python tools/k1_insitu.py "" "SPLIT=1" "SPLIT=2" "SPLIT=4" 2>/dev/null | grep -v "^# setup"
echo "# --- dim 256 (1-KiB lines), 24 M rows, 16 GiB cache, fan-out 15,10,5"
ROWS=24000000 DIM=256 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 REPS=2 timeout -k 10 900 python tools/k1_insitu.py "" "SPLIT=1" "SPLIT=2" 2>/dev/null | grep -v "^# setup") > gpurun_out/r03/k1_split.txt 2>&1
cat gpurun_out/r03/k1_split.txt
