#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 REPS=3 timeout -k 10 900 python tools/k1_insitu.py "" "GRID=2048" "WAVES=1" "WAVES=4" "GRID=4096" > gpurun_out/r03/k1_insitu_papers5.txt 2> gpurun_out/r03/k1_insitu_papers5.err; echo "insitu papers 72k rc=$?"; cat gpurun_out/r03/k1_insitu_papers5.txt
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12.75 REPS=3 timeout -k 10 900 python tools/k1_insitu.py "" "WAVES=1" "WAVES=4" "GRID=8192" > gpurun_out/r03/k1_insitu_papers6.txt 2> gpurun_out/r03/k1_insitu_papers6.err; echo "insitu papers 315k rc=$?"; cat gpurun_out/r03/k1_insitu_papers6.txt
REPS=3 timeout -k 10 600 python tools/k1_insitu.py "" "WAVES=1" "WAVES=4" "GRID=2048" > gpurun_out/r03/k1_insitu_default2.txt 2> gpurun_out/r03/k1_insitu_default2.err; echo "insitu default rc=$?"; cat gpurun_out/r03/k1_insitu_default2.txt
