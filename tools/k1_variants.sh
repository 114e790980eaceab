#!/bin/bash
# GPU box: the A/B table of the probe+gather kernel kept under profiles/ (r02_k1_variants.txt):
#   1. tools/k1_bench (micro-benchmark, same batch repeated: hot tag sets): product kernel P5 = FULL variant, P4 = predicated variant,
#      P6 = FULL without miss bookkeeping, against the experimental variants incl. rows staged through LDS by LDS-DMA
#      (the reason the product keeps register staging), at 100 % and 32 % hits;
#   2. tools/k1_insitu.py (the bench workload itself, every launch behind a PCIe-bound cold fill): launch geometry sweep + the all-hit leg.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_k1_variants.txt
cd /tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include $R/tools/k1_bench.hip $R/coala-gnn_amd/csrc/coala_host.cpp -o /tmp/k1_bench -lrt
{
echo "# tools/k1_bench 2000000 36864 1024 <hit%>   (separate hipEvent brackets: every figure includes ~4.6 us of bracket; 30 interleaved repetitions)"
for hit in 100 32; do /tmp/k1_bench 2000000 36864 1024 $hit; done
echo
echo "# tools/k1_insitu.py (default bench workload; one cache handle per variant; development build with launch-geometry knobs; events attached to the launches: no bracket overhead)"
REPS=1 ALLHIT=1 python3 $R/tools/k1_insitu.py "GRID=1024" "GRID=2048" "GRID=4096" "GRID=8192" "GRID=16384" "GRID=8192,WAVES=1" "GRID=4096,WAVES=4" "GRID=8192,PASSES=2" 2>&1 | grep -v amdgpu.ids
} > $OUT 2>&1
cat $OUT
