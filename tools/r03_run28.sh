#!/bin/bash
# the cold fill's grid under a training load: does a wider grid win back the 5 % the fill is stretched by beside the consumer's kernels?  (development build, COALA_K2_GRID)
set -o pipefail
mkdir -p gpurun_out/r03
L=$PWD/coala-gnn_amd/lib/libcoala_hip_dev.so
(for g in 16 20 24 32 16; do
  echo "=== COALA_K2_GRID=$g"
  COALA_K2_GRID=$g COALA_HIP_LIB=$L STEPS=1600 timeout -k 10 300 python tools/fetch_gap_probe.py 2>&1 | grep -v amdgpu.ids | head -2
done) > gpurun_out/r03/k2_grid_under_load.txt 2>&1
cat gpurun_out/r03/k2_grid_under_load.txt
