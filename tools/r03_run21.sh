#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 bash tools/k1_sq_counters.sh papers100m > gpurun_out/r03/k1_sq_papers100m.txt 2>&1; cat gpurun_out/r03/k1_sq_papers100m.txt
bash tools/k1_sq_counters.sh default > gpurun_out/r03/k1_sq_default.txt 2>&1; cat gpurun_out/r03/k1_sq_default.txt
tail -3 gpurun_out/sq_papers100m_1.log
