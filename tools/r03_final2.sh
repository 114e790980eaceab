#!/bin/bash
# wave-scheduler counters of the final K1 (both shapes), then tools/r03_final.sh
set -o pipefail
mkdir -p gpurun_out/r03
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 bash tools/k1_sq_counters.sh papers100m 2>&1 | grep -A31 "^--- K1" | head -32 > gpurun_out/r03/k1_sq_papers100m_final.txt
bash tools/k1_sq_counters.sh default 2>&1 | grep -A31 "^--- K1" | head -32 > gpurun_out/r03/k1_sq_default_final.txt
grep "wave lifetime\|parked\|instructions per wave" gpurun_out/r03/k1_sq_*_final.txt
bash tools/r03_final.sh
