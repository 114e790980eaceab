#!/bin/bash
# GPU box: rocprofv3 kernel trace of tools/k1_insitu.py --stages (K1's dependency chain cut after each link, launched behind a real
# step's cold fill) -> per-kernel average durations, next to the HIP-event numbers the tool prints itself.
# the profiler's preloaded library starts the HIP runtime before python does: bench.py's os.environ.setdefault comes too late there,
# so the queue count it reports has to be exported by the shell that starts the profiler
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${OUT:-$R/gpurun_out/r02_k1_fixed_cost.txt}   # shape from the environment of tools/k1_insitu.py (ROWS DIM FANOUT CACHE_MB DEG)
export TMPDIR=/tmp
cd /tmp
python3 $R/coala-gnn_amd/build.py --dev > /dev/null
d=$R/gpurun_out/prof_k1_stages
rm -rf $d
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/k1_insitu.py --stages > $OUT 2>&1
echo "--- rocprofv3 --kernel-trace --stats of the same run (Name, Calls, AverageNs, MinNs, MaxNs)" >> $OUT
f=$(find $d -name "*kernel_stats.csv" | head -1)
python3 - "$f" >> $OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k1_empty_kernel" in n or "probe_gather_kernel" in n or "miss_fill_kernel" in n:
        short = n.replace("(anonymous namespace)::", "").split("(")[0]
        print(f"{short[:72]:72s} calls {r['Calls']:>6s}  avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f} us  max {float(r['MaxNs'])/1e3:8.2f} us")
PY
rm -rf $d
grep -v amdgpu.ids $OUT
