#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 passes over the default bench command, results under gpurun_out/prof_*.
#   kernel trace + stats  -> per-kernel durations (must agree with bench.py's hipEvent numbers)
#   --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md: 3 + 2 TCC slots, not in one pass)
# tools/summarize_profiles.py then writes the tracked summaries into profiles/.
# the profiler's preloaded library starts the HIP runtime before python does: bench.py's os.environ.setdefault comes too late there,
# so the queue count it reports has to be exported by the shell that starts the profiler
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
ARGS="${BENCH_ARGS:---no-cpu-baseline --no-allhit --epoch-steps 0 --no-fanout-leg --no-color-affinity-leg}"   # the timed region only: the extra legs would add launches of the same kernels after it
TAG="${TAG:-default}"
export TMPDIR=/tmp
cd /tmp
mkdir -p $R/gpurun_out/prof_${TAG}_trace $R/gpurun_out/prof_${TAG}_fetch $R/gpurun_out/prof_${TAG}_write
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_trace -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_trace.log 2>&1
echo "trace pass done"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
echo "fetch pass done"
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${TAG}_write.log 2>&1
echo "write pass done"
cd $R
python3 tools/summarize_profiles.py $TAG
