#!/bin/bash
# GPU suite, then the rocprof / PMC passes and bench lines (tools/r03_final.sh), then the wave-scheduler counters, at HEAD
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputest_final.log 2>&1; echo "gputests rc=$?"; tail -4 gpurun_out/r03/gputest_final.log
bash tools/r03_final.sh
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 bash tools/k1_sq_counters.sh papers100m 2>&1 | grep -A31 "^--- K1" | head -32 > gpurun_out/r03/k1_sq_papers100m_final.txt
bash tools/k1_sq_counters.sh default 2>&1 | grep -A31 "^--- K1" | head -32 > gpurun_out/r03/k1_sq_default_final.txt
grep "wave lifetime\|parked\|instructions per wave" gpurun_out/r03/k1_sq_*_final.txt
