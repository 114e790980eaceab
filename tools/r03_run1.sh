#!/bin/bash
# round 3, GPU call 1: host facts, the reference-API cold tier (shm) beside hipHostMalloc, the whole GPU suite after the comm changes
set -o pipefail
mkdir -p gpurun_out/r03
python tools/host_probe.py --gb 1,8,41 > gpurun_out/r03/host_probe.json 2> gpurun_out/r03/host_probe.err
echo "host_probe rc=$?"
X="--steps 100 --epoch-steps 0 --no-fanout-leg --no-color-affinity-leg --no-cpu-baseline"
python bench.py $X > gpurun_out/r03/bench_host.json 2> gpurun_out/r03/bench_host.err && echo bench_host ok
python bench.py $X --cold-tier shm > gpurun_out/r03/bench_shm.json 2> gpurun_out/r03/bench_shm.err && echo bench_shm ok
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03/gpu_tests.log 2>&1
echo "tests rc=$?"
tail -3 gpurun_out/r03/gpu_tests.log
