#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
X="--steps 200 --epoch-steps 0 --no-fanout-leg --no-color-affinity-leg --no-cpu-baseline"
python bench.py $X --cold-tier shm > gpurun_out/r03/bench_shm.json 2> gpurun_out/r03/bench_shm.err && echo bench_shm ok
bash tools/numa_probe.sh > gpurun_out/r03/numa_near_far.txt 2>&1; echo "numa rc=$?"; cat gpurun_out/r03/numa_near_far.txt
timeout -k 10 600 python -m pytest tests/test_dist_gpu.py -k "IGB-large" -x -q > gpurun_out/r03/cfg4_test.log 2>&1; echo "cfg4 test rc=$?"; tail -3 gpurun_out/r03/cfg4_test.log
timeout -k 10 600 python tools/dist_config_probe.py > gpurun_out/r03/igb_large_scaled.json 2> gpurun_out/r03/igb_large_scaled.err; echo "cfg4 probe rc=$?"; tail -2 gpurun_out/r03/igb_large_scaled.err
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not IGB-large" > gpurun_out/r03/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03/gpu_tests.log
