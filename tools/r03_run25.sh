#!/bin/bash
# like for like: the round-2/3 probe (libcoala_hip_old.so = commit 52aa5d2's coala_cache.hip) against the lane-parallel probe (development build), same harnesses,
# output buffers in rotation everywhere
set -o pipefail
mkdir -p gpurun_out/r03
L=$PWD/coala-gnn_amd/lib
(for lib in libcoala_hip_old.so libcoala_hip_dev.so; do
  echo "=== $lib: hit sweep (three output buffers in rotation)"
  COALA_HIP_LIB=$L/$lib HITS=32,75,100 timeout -k 10 600 python tools/k1_dim_sweep.py 2>/dev/null | grep "^dim"
  echo "=== $lib: default workload in situ + all-hit leg (three output buffers)"
  K1_LIB=$L/$lib REPS=1 ALLHIT=1 timeout -k 10 600 python tools/k1_insitu.py "" 2>/dev/null | grep -v "^# setup"
  echo "=== $lib: the same with ONE output buffer"
  K1_LIB=$L/$lib OUT_BUFFERS=1 REPS=1 ALLHIT=1 timeout -k 10 600 python tools/k1_insitu.py "" 2>/dev/null | grep -v "^# setup"
done) > gpurun_out/r03/k1_old_vs_new.txt 2>&1
cat gpurun_out/r03/k1_old_vs_new.txt
