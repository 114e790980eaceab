#!/bin/bash
# GPU box: everything under profiles/ for one round.  PART=1: the rocprofv3 passes over bench.py (default, all-hit, configs[3] shape);
# PART=2: the probes; unset: both (more than one gpurun call's time limit since round 3).  Summaries land in gpurun_out/profiles_out/.
# the profiler's preloaded library starts the HIP runtime before python does: bench.py's os.environ.setdefault comes too late there,
# so the queue count it reports has to be exported by the shell that starts the profiler
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
R=${GRAFT_REPO_ROOT:-$PWD}
export ROUND=${ROUND:-r04} TMPDIR=/tmp
O=$R/gpurun_out/profiles_out
mkdir -p $O
cd $R
if [ "${PART:-1}" = "1" ] || [ -z "$PART" ]; then
bash tools/profile_round.sh > $R/gpurun_out/profile_default.log 2>&1; echo "default profile rc=$?"
TAG=allhit BENCH_ARGS="--rows 2000000 --mode allhit --prewarm 3 --steps 100 --no-cpu-baseline --epoch-steps 0 --no-fanout-leg --no-color-affinity-leg" bash tools/profile_round.sh > $R/gpurun_out/profile_allhit.log 2>&1; echo "allhit profile rc=$?"
# BASELINE configs[3]'s single-GPU shape (512-B lines, 16 GiB cache): kernel trace + the two PMC passes for probe_gather_kernel<128, ...>
TAG=papers100m BENCH_ARGS="--rows 111059956 --dim 128 --fanout 15,10,5 --cache-mb 16384 --no-cpu-baseline --no-allhit --epoch-steps 0 --no-fanout-leg --no-color-affinity-leg" bash tools/profile_round.sh > $R/gpurun_out/profile_papers100m.log 2>&1; echo "papers100m profile rc=$?"
fi
if [ "${PART:-2}" = "2" ] || [ -z "$PART" ]; then
bash tools/dist_profile.sh > /dev/null 2>&1; grep -v amdgpu.ids $R/gpurun_out/r02_dist_step_kernels.txt > $O/${ROUND}_dist_step_kernels.txt; echo "dist rc=$?"
COALA_K1_GRID=8192 bash tools/k1_stages_profile.sh > /dev/null 2>&1; grep -v "amdgpu.ids\|^[EW]2026" $R/gpurun_out/r02_k1_fixed_cost.txt > $O/${ROUND}_k1_fixed_cost.txt; echo "k1 stages rc=$?"
python3 tools/k1_dim_sweep.py 2>&1 | grep -v amdgpu.ids > $O/${ROUND}_k1_hit_sweep.txt; echo "dim sweep rc=$?"
(for g in 1024 4096 16384 65536; do COALA_K1_GRID=$g python3 tools/k1_big_batch_grid.py 2>&1 | grep "^GRID"; done) > $O/${ROUND}_k1_big_batches.body; echo "k1 big batches rc=$?"
bash tools/dist_overlap_profile.sh > /dev/null 2>&1; cp $R/gpurun_out/r02_dist_overlap.txt $O/${ROUND}_dist_overlap.body; echo "dist overlap rc=$?"
python3 tools/fetch_gap_probe.py 2>&1 | grep -v amdgpu.ids > $O/${ROUND}_prefetch_fetch_gaps.txt; echo "gap probe rc=$?"
(echo "# tools/sampler_fanout_probe.py (MI355X; 10 M-node power-law graph, 1024 seeds; round 1: 0.107 / 0.110 / 0.210 / 0.197 ms for 5,5 / 10,10 / 15,10,5 / 10,10,10)"; python3 tools/sampler_fanout_probe.py 2>&1 | grep -v amdgpu.ids) > $O/${ROUND}_sampler_fanouts.txt; echo "sampler rc=$?"
python3 tools/backend_compare_probe.py 2>/dev/null | grep "^{" | python3 -m json.tool > $O/${ROUND}_backend_compare.json; echo "backend compare rc=$?"
for g in community powerlaw; do python3 tools/color_affinity_probe.py --graph $g 2>/dev/null | grep "^{" | python3 -m json.tool > $O/${ROUND}_color_affinity_$g.json; echo "colour $g rc=$?"; done
(echo "# --- compaction off (K2_SPARSE=0, development build): every tile walked chunk by chunk, the round-1 behaviour"; K2_SPARSE=0 python3 tools/k2_sparse_probe.py 2>&1 | grep "^miss"
 echo "# --- product (tiles with <= 32 of 64 rows missing are ranked at once and streamed compacted)"; python3 tools/k2_sparse_probe.py 2>&1 | grep "^miss") > $O/${ROUND}_k2_sparse_misses.body; echo "k2 sparse rc=$?"
fi
ls -la $O
