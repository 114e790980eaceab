#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 REPS=3 timeout -k 10 900 python tools/k1_insitu.py "" "SINGLE=1" > gpurun_out/r03/k1_single_72k.txt 2> gpurun_out/r03/k1_single_72k.err; echo "72k rc=$?"; cat gpurun_out/r03/k1_single_72k.txt
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12.75 REPS=3 timeout -k 10 900 python tools/k1_insitu.py "" "SINGLE=1" > gpurun_out/r03/k1_single_315k.txt 2> gpurun_out/r03/k1_single_315k.err; echo "315k rc=$?"; cat gpurun_out/r03/k1_single_315k.txt
REPS=3 ALLHIT=1 timeout -k 10 600 python tools/k1_insitu.py "" "SINGLE=1" > gpurun_out/r03/k1_single_default.txt 2> gpurun_out/r03/k1_single_default.err; echo "default rc=$?"; cat gpurun_out/r03/k1_single_default.txt
