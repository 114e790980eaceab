#!/bin/bash
# does a raised wave priority (s_setprio 3) in the cold fill protect it from the consumer's kernels?  fetch duration under a training load,
# development build against the same build with -DK2_SETPRIO=3 (both through the ctypes table), alternating
set -o pipefail
mkdir -p gpurun_out/r03
L=coala-gnn_amd/lib
for rep in 1 2; do
for lib in libcoala_hip_dev.so libcoala_hip_prio.so; do
  echo "=== $lib (rep $rep)"
  COALA_HIP_LIB=$PWD/$L/$lib STEPS=1600 timeout -k 10 300 python tools/fetch_gap_probe.py 2>&1 | grep -v amdgpu.ids | head -3
done
done > gpurun_out/r03/k2_setprio.txt 2>&1
cat gpurun_out/r03/k2_setprio.txt
