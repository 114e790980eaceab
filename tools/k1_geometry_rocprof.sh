#!/bin/bash
# GPU box: rocprofv3 durations of probe_gather_kernel in situ per launch geometry (tools/k1_insitu.py variants share one kernel
# symbol: grouped by grid / workgroup size from the kernel trace, last 200 launches of each group = the timed minibatches).
# the profiler's preloaded library starts the HIP runtime before python does: bench.py's os.environ.setdefault comes too late there,
# so the queue count it reports has to be exported by the shell that starts the profiler
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp REPS=1
cd /tmp
d=$R/gpurun_out/prof_k1geo
rm -rf $d
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $R/tools/k1_insitu.py "$@" > $R/gpurun_out/k1geo.log 2>&1
f=$(find $d -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "probe_gather_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = 620   # tools/k1_insitu.py: 420 warm-up + 200 timed minibatches per variant
for i in range(0, len(rows), per):
    grp = rows[i: i + per]
    last = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp[-200:]]
    wg = grp[-1].get("Workgroup_Size_X") or grp[-1].get("Workgroup_Size")
    gs = grp[-1].get("Grid_Size_X") or grp[-1].get("Grid_Size")
    print(f"variant {i // per}: workgroup {wg:>4s}, grid of the last launch {gs:>8s} threads: {len(grp)} launches, last {len(last)}: avg {sum(last)/len(last)/1e3:6.2f} us  min {min(last)/1e3:6.2f} us")
PY
rm -rf $d
grep "K1 " $R/gpurun_out/k1geo.log
