#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "test_cache_gpu or test_fuzz or test_golden or test_loader_gpu or rehearsal or ring_exhaustion or split_phase" > gpurun_out/r03/gpu_tests5.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r03/gpu_tests5.log
(echo "# --- compaction off (K2_SPARSE=0, development build)"; K2_SPARSE=0 python3 tools/k2_sparse_probe.py 2>&1 | grep "^miss"
 echo "# --- product"; python3 tools/k2_sparse_probe.py 2>&1 | grep "^miss") > gpurun_out/r03/k2_sparse.txt; echo "k2 sparse rc=$?"; cat gpurun_out/r03/k2_sparse.txt
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 REPS=1 timeout -k 10 900 python tools/k1_insitu.py "" "GRID=512" "GRID=1024" "GRID=2048" "WAVES=1" "WAVES=4" > gpurun_out/r03/k1_insitu_papers3.txt 2> gpurun_out/r03/k1_insitu_papers3.err; echo "insitu papers 72k rc=$?"; cat gpurun_out/r03/k1_insitu_papers3.txt
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12.75 REPS=1 timeout -k 10 900 python tools/k1_insitu.py "" "GRID=2048" "GRID=4096" "GRID=8192" > gpurun_out/r03/k1_insitu_papers4.txt 2> gpurun_out/r03/k1_insitu_papers4.err; echo "insitu papers 315k rc=$?"; cat gpurun_out/r03/k1_insitu_papers4.txt
python bench.py --epoch-steps 0 --no-fanout-leg --no-color-affinity-leg > gpurun_out/r03/bench_default_noepoch.json 2> gpurun_out/r03/bench_default_noepoch.err; echo "bench rc=$?"
python tools/shm_two_process_probe.py > gpurun_out/r03/bench_shm_second_mapping.json 2> gpurun_out/r03/bench_shm_second_mapping.err; echo "shm 2proc rc=$?"; tail -3 gpurun_out/r03/bench_shm_second_mapping.err
