#!/bin/bash
# the cold fill's grid WITHOUT a consumer beside it (default workload in situ: K2 us and step ms), 16 / 20 / 24 / 32 blocks
set -o pipefail
mkdir -p gpurun_out/r03
(for g in 16 20 24 32 16 24; do
  echo "=== COALA_K2_GRID=$g"
  COALA_K2_GRID=$g REPS=1 timeout -k 10 300 python tools/k1_insitu.py "" 2>/dev/null | grep -v "^# setup"
done) > gpurun_out/r03/k2_grid_alone.txt 2>&1
cat gpurun_out/r03/k2_grid_alone.txt
