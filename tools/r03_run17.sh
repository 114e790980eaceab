#!/bin/bash
# cross-wave prefetch in K1 (COALA_K1_PF = distance in blocks) against the product kernel, in situ, three output buffers in rotation
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests/test_fuzz_gpu.py -x -q -k "other_launch_shapes and (11 or 12)" > gpurun_out/r03/fuzz_pf.log 2>&1; echo "fuzz pf rc=$?"; tail -3 gpurun_out/r03/fuzz_pf.log
(echo "# --- configs[3] shape, avg degree 12 (~289 k rows)"
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 REPS=2 timeout -k 10 900 python tools/k1_insitu.py "" "PF=4096" "PF=8192" "SINGLE=1" "SINGLE=1,PF=4096" "SINGLE=1,PF=6144" "SINGLE=1,PF=8192" 2>/dev/null | grep -v "^# setup"
echo "# --- configs[3] shape, avg degree 6 (~72 k rows)"
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 REPS=2 timeout -k 10 900 python tools/k1_insitu.py "" "PF=2048" "SINGLE=1,PF=2048" "SINGLE=1,PF=4096" 2>/dev/null | grep -v "^# setup") > gpurun_out/r03/k1_pf.txt 2>&1
cat gpurun_out/r03/k1_pf.txt
