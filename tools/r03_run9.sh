#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "(test_dist_gpu or test_fuzz) and not IGB-large and not papers100M" > gpurun_out/r03/gpu_tests9.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03/gpu_tests9.log
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12.75 timeout -k 10 600 python tools/k1_insitu.py --stages > gpurun_out/r03/k1_stages_papers_315k.txt 2> gpurun_out/r03/k1_stages_papers_315k.err; echo "stages 315k rc=$?"; cat gpurun_out/r03/k1_stages_papers_315k.txt
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 timeout -k 10 600 python tools/k1_insitu.py --stages > gpurun_out/r03/k1_stages_papers_72k.txt 2> gpurun_out/r03/k1_stages_papers_72k.err; echo "stages 72k rc=$?"; cat gpurun_out/r03/k1_stages_papers_72k.txt
timeout -k 10 600 python tools/k1_insitu.py --stages > gpurun_out/r03/k1_stages_default.txt 2> gpurun_out/r03/k1_stages_default.err; echo "stages default rc=$?"; cat gpurun_out/r03/k1_stages_default.txt
python tools/dist_breakdown.py --mode both 2>&1 | grep -v amdgpu.ids > gpurun_out/r03/dist_breakdown.txt; cat gpurun_out/r03/dist_breakdown.txt
timeout -k 10 600 python tools/dist_config_probe.py > gpurun_out/r03/igb_large_scaled.json 2> gpurun_out/r03/igb_large_scaled.err; echo "cfg4 probe rc=$?"; tail -2 gpurun_out/r03/igb_large_scaled.err
