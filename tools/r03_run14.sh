#!/bin/bash
# K1 in situ with the output rotating over 3 buffers (as bench.py and the loaders see it) against one reused buffer; plain against nontemporal row stores
set -o pipefail
mkdir -p gpurun_out/r03
L=$PWD/coala-gnn_amd/lib
(for lib in libcoala_hip_dev.so libcoala_hip_ntst.so; do for nb in 1 3; do
  echo "=== $lib OUT_BUFFERS=$nb  configs[3] shape"
  K1_LIB=$L/$lib OUT_BUFFERS=$nb ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 REPS=2 timeout -k 10 600 python tools/k1_insitu.py "" 2>/dev/null | grep -v "^# setup"
done; done
for lib in libcoala_hip_dev.so libcoala_hip_ntst.so; do for nb in 1 3; do
  echo "=== $lib OUT_BUFFERS=$nb  default workload"
  K1_LIB=$L/$lib OUT_BUFFERS=$nb REPS=2 ALLHIT=1 timeout -k 10 600 python tools/k1_insitu.py "" 2>/dev/null | grep -v "^# setup"
done; done) > gpurun_out/r03/k1_out_buffers.txt 2>&1
cat gpurun_out/r03/k1_out_buffers.txt
bash tools/r03_run13.sh
