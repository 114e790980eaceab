#!/bin/bash
# old probe (libcoala_hip_old.so) / lane-parallel probe with the loop's lane-derived values no longer hoisted (development build) / the same with 5 waves per SIMD
# demanded of the 4-KiB-line kernels (libcoala_hip_var.so): the low-hit-ratio points of the sweep and the in-situ workloads, output buffers in rotation
set -o pipefail
mkdir -p gpurun_out/r03
L=$PWD/coala-gnn_amd/lib
(for lib in libcoala_hip_old.so libcoala_hip_dev.so libcoala_hip_var.so; do
  echo "=== $lib: hit sweep"
  COALA_HIP_LIB=$L/$lib HITS=0,32,75 SHAPES=512:123904:4000000,1024:36864:2000000,1024:123904:2000000 timeout -k 10 600 python tools/k1_dim_sweep.py 2>/dev/null | grep "^dim"
  echo "=== $lib: default workload in situ + all-hit leg"
  K1_LIB=$L/$lib REPS=1 ALLHIT=1 timeout -k 10 600 python tools/k1_insitu.py "" 2>/dev/null | grep -v "^# setup"
done
for lib in libcoala_hip_old.so libcoala_hip_dev.so; do
  echo "=== $lib: configs[3] shape ~289 k rows, ~72 k rows"
  K1_LIB=$L/$lib ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 REPS=1 timeout -k 10 600 python tools/k1_insitu.py "" 2>/dev/null | grep -v "^# setup"
  K1_LIB=$L/$lib ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 REPS=1 timeout -k 10 600 python tools/k1_insitu.py "" 2>/dev/null | grep -v "^# setup"
done) > gpurun_out/r03/k1_old_vs_new2.txt 2>&1
cat gpurun_out/r03/k1_old_vs_new2.txt
