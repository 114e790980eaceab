#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not IGB-large and not papers100M and not test_bench" > gpurun_out/r03/gpu_tests3.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03/gpu_tests3.log
# K1 in situ, configs[3] shape (512-B lines, 16 GiB cache)
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 REPS=1 timeout -k 10 900 python tools/k1_insitu.py "" "PASSES=4" "PASSES=16" "TAG64=1,PASSES=4" "TAG64=1,PASSES=8" "GRID=2048" "GRID=4096" "GRID=2048,PASSES=16" "WAVES=4" > gpurun_out/r03/k1_insitu_papers.txt 2> gpurun_out/r03/k1_insitu_papers.err; echo "insitu papers rc=$?"; cat gpurun_out/r03/k1_insitu_papers.txt
REPS=1 ALLHIT=1 timeout -k 10 600 python tools/k1_insitu.py "" "TAG64=1" "PASSES=8" > gpurun_out/r03/k1_insitu_default.txt 2> gpurun_out/r03/k1_insitu_default.err; echo "insitu default rc=$?"; cat gpurun_out/r03/k1_insitu_default.txt
TIER=shm bash tools/numa_probe.sh > gpurun_out/r03/numa_shm_near_far.txt 2>&1; echo "numa rc=$?"; cat gpurun_out/r03/numa_shm_near_far.txt
