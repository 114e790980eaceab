#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03
tools/tlb_probe > gpurun_out/r03/tlb_probe.txt 2>&1; echo "tlb rc=$?"; cat gpurun_out/r03/tlb_probe.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not IGB-large and not papers100M and not IGB-medium and not (test_bench and not rehearsal) and not color_affinity and not backend_compare" > gpurun_out/r03/gpu_tests4.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r03/gpu_tests4.log
(echo "# --- compaction off (K2_SPARSE=0, development build)"; K2_SPARSE=0 python3 tools/k2_sparse_probe.py 2>&1 | grep "^miss"
 echo "# --- product"; python3 tools/k2_sparse_probe.py 2>&1 | grep "^miss") > gpurun_out/r03/k2_sparse.txt; echo "k2 sparse rc=$?"; cat gpurun_out/r03/k2_sparse.txt
# K1 against the cache size at 512-B lines: the same batches on a 4 GiB and a 16 GiB cache
for mb in 4096 16384; do CACHE_MB=$mb SHAPES=128:73728:40000000,128:294912:40000000,128:1081344:40000000 HITS=0,50,62,100 python3 tools/k1_dim_sweep.py 2>&1 | grep "^dim\|^#" | sed "s/^/cache ${mb} MB: /"; done > gpurun_out/r03/k1_cache_size.txt; cat gpurun_out/r03/k1_cache_size.txt
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12.75 REPS=1 timeout -k 10 900 python tools/k1_insitu.py "" "PASSES=4" "TAG64=1,PASSES=4" "PASSES=2" > gpurun_out/r03/k1_insitu_papers2.txt 2> gpurun_out/r03/k1_insitu_papers2.err; echo "insitu papers rc=$?"; cat gpurun_out/r03/k1_insitu_papers2.txt
timeout -k 10 600 python tools/dist_config_probe.py > gpurun_out/r03/igb_large_scaled.json 2> gpurun_out/r03/igb_large_scaled.err; echo "cfg4 probe rc=$?"; tail -2 gpurun_out/r03/igb_large_scaled.err
