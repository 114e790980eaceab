#!/bin/bash
# GPU box: the prefetching loader under a training load -- tools/prefetch_probe.py (host-side split: consumer wait, producer
# stages) run under rocprofv3 --kernel-trace --stats, so that the durations of the cache kernels in THAT context (cold fill beside
# the training kernels) sit next to the host numbers.  Output: gpurun_out/r02_prefetch_epoch.txt
# the profiler's preloaded library starts the HIP runtime before python does: bench.py's os.environ.setdefault comes too late there,
# so the queue count it reports has to be exported by the shell that starts the profiler
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02_prefetch_epoch.txt
export TMPDIR=/tmp
cd /tmp
d=$R/gpurun_out/prof_prefetch
rm -rf $d
timeout -k 10 800 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/prefetch_probe.py > $OUT 2>&1
echo "--- rocprofv3 --kernel-trace --stats of the same run: cache / sampler kernels, then the largest others" >> $OUT
f=$(find $d -name "*kernel_stats.csv" | head -1)
python3 - "$f" >> $OUT <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ours = [r for r in rows if "anonymous namespace" in r["Name"]]
others = [r for r in rows if r not in ours][:8]
for r in ours + others:
    short = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-70:]
    print(f"{short:70s} calls {r['Calls']:>7s}  avg {float(r['AverageNs'])/1e3:9.2f} us  total {float(r['TotalDurationNs'])/1e6:9.1f} ms  {r['Percentage']:>6s} %")
PY
rm -rf $d
grep -v "amdgpu.ids\|^[EW]2026" $OUT
