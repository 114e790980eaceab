#!/bin/bash
# XCD-affine K1 against the product K1, in situ on the configs[3] shape (512-B lines, 16 GiB cache) and on the default workload
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests/test_fuzz_gpu.py -x -q -k "other_launch_shapes and (11 or 12)" > gpurun_out/r03/fuzz_xcd.log 2>&1; echo "fuzz xcd rc=$?"; tail -3 gpurun_out/r03/fuzz_xcd.log
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12.75 SEGMENTS=1 REPS=3 timeout -k 10 900 python tools/k1_insitu.py "" "XCD=1" "XCD=2" > gpurun_out/r03/k1_xcd_315k.txt 2> gpurun_out/r03/k1_xcd_315k.err; echo "315k rc=$?"; cat gpurun_out/r03/k1_xcd_315k.txt
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 SEGMENTS=1 REPS=3 timeout -k 10 900 python tools/k1_insitu.py "" "XCD=1" "XCD=2" > gpurun_out/r03/k1_xcd_72k.txt 2> gpurun_out/r03/k1_xcd_72k.err; echo "72k rc=$?"; cat gpurun_out/r03/k1_xcd_72k.txt
REPS=2 ALLHIT=1 timeout -k 10 600 python tools/k1_insitu.py "" "XCD=1" > gpurun_out/r03/k1_xcd_default.txt 2> gpurun_out/r03/k1_xcd_default.err; echo "default rc=$?"; cat gpurun_out/r03/k1_xcd_default.txt
