#!/bin/bash
# GPU box: device budget + measured HBM bytes (rocprofv3 PMC, separate passes) of one rank's share of an 8-way step,
# round-1 sequence ("before") against the split-phase one ("after").  Output: gpurun_out/r02_dist_step_kernels.txt
# the profiler's preloaded library starts the HIP runtime before python does: bench.py's os.environ.setdefault comes too late there,
# so the queue count it reports has to be exported by the shell that starts the profiler
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_dist_step_kernels.txt
export TMPDIR=/tmp
cd /tmp
python3 $R/tools/dist_breakdown.py --mode both > $OUT 2>&1
for mode in before after bucketed; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$R/gpurun_out/prof_dist_${mode}_${c}
    rm -rf $d
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/tools/dist_breakdown.py --mode $mode --reps 20 --pmc > /dev/null 2>&1
    echo "--- rocprofv3 --pmc $c, mode $mode (per step; 21 steps incl. the warm-up serve -> divided by 20: slight over-count)" >> $OUT
    python3 $R/tools/pmc_by_kernel.py $d $c 20 "anonymous namespace|rocclr_copyBuffer" 200 >> $OUT
    rm -rf $d
  done
done
cat $OUT
