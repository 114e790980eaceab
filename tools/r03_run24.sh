#!/bin/bash
# K1 with the wave index made provably uniform (readfirstlane) against the same kernel without (libcoala_hip_var.so, -DK1_NO_UNIFORM_WAVE), in situ
set -o pipefail
mkdir -p gpurun_out/r03
L=$PWD/coala-gnn_amd/lib
(for rep in 1 2; do for lib in libcoala_hip_var.so libcoala_hip_dev.so; do
  echo "=== $lib  configs[3] shape, ~289 k rows / ~72 k rows / default workload"
  K1_LIB=$L/$lib ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 REPS=1 timeout -k 10 600 python tools/k1_insitu.py "" 2>/dev/null | grep -v "^# setup"
  K1_LIB=$L/$lib ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 REPS=1 timeout -k 10 600 python tools/k1_insitu.py "" 2>/dev/null | grep -v "^# setup"
  K1_LIB=$L/$lib REPS=1 ALLHIT=1 timeout -k 10 600 python tools/k1_insitu.py "" 2>/dev/null | grep -v "^# setup"
done; done) > gpurun_out/r03/k1_uniform_wave.txt 2>&1
cat gpurun_out/r03/k1_uniform_wave.txt
