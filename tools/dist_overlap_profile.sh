#!/bin/bash
# GPU box: kernel trace of the in-process distributed fetch with a real miss ratio -> which row copies (comm stream) ran beside
# which cold fills (caller's stream).  Output: gpurun_out/r02_dist_overlap.txt
# the profiler's preloaded library starts the HIP runtime before python does: bench.py's os.environ.setdefault comes too late there,
# so the queue count it reports has to be exported by the shell that starts the profiler
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_dist_overlap.txt
export TMPDIR=/tmp
cd /tmp
d=$R/gpurun_out/prof_dist_overlap
rm -rf $d
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $R/tools/dist_overlap_trace.py --ranks ${RANKS:-2} --steps 12 --rounds ${ROUNDS:-4} 2>&1 | grep "^#" > $OUT
f=$(find $d -name "*kernel_trace.csv" | head -1)
python3 $R/tools/dist_overlap_trace.py --summarize $f --last 16 >> $OUT
rm -rf $d
cat $OUT
