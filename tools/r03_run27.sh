#!/bin/bash
# the loop-free one-wave-per-chunk K1 (COALA_K1_SINGLE=1, development build) against the looping product kernel with the lane-parallel probe, lines of 1 KiB and more
set -o pipefail
mkdir -p gpurun_out/r03
L=$PWD/coala-gnn_amd/lib/libcoala_hip_dev.so
(for single in 0 1; do
  echo "=== COALA_K1_SINGLE=$single: hit sweep (three output buffers in rotation)"
  COALA_K1_SINGLE=$single COALA_HIP_LIB=$L HITS=0,32,75,100 SHAPES=256:262144:4000000,512:123904:4000000,1024:36864:2000000,1024:123904:2000000 timeout -k 10 600 python tools/k1_dim_sweep.py 2>/dev/null | grep "^dim"
done
echo "=== default workload in situ + all-hit leg (three output buffers), two handles each"
REPS=2 ALLHIT=1 timeout -k 10 600 python tools/k1_insitu.py "" "SINGLE=1" 2>/dev/null | grep -v "^# setup") > gpurun_out/r03/k1_single_lane_parallel.txt 2>&1
cat gpurun_out/r03/k1_single_lane_parallel.txt
