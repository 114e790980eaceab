#!/bin/bash
# the redirecting K1 without a row map (destination = position - begin, computed at store time): parity, then one rank's share of an 8-way step
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests/test_fuzz_gpu.py tests/test_dist_gpu.py tests/test_cache_gpu.py -x -q > gpurun_out/r03/parity_redir.log 2>&1; echo "parity rc=$?"; tail -3 gpurun_out/r03/parity_redir.log
bash tools/dist_profile.sh > /dev/null 2>&1; grep -v amdgpu.ids gpurun_out/r02_dist_step_kernels.txt | head -13
