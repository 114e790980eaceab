#!/bin/bash
# per-kernel durations of the two-kernel K1 (rocprofv3 kernel trace)
set -o pipefail
R=$PWD
mkdir -p gpurun_out/r03 gpurun_out/prof_split
export TMPDIR=/tmp ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 REPS=1 GPU_MAX_HW_QUEUES=8
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_split -- python3 $R/tools/k1_insitu.py "SPLIT=1" "SPLIT=2" "" > $R/gpurun_out/r03/k1_split_trace.log 2>&1
echo "rc=$?"
cd $R
f=$(find gpurun_out/prof_split -name "*kernel_stats.csv" | head -1)
echo $f
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "probe_gather" in n or "hit_copy" in n or "miss_fill" in n:
        print(f'{n[:110]:110s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"]) / 1e3:8.2f} us  min {float(r["MinNs"]) / 1e3:8.2f}  max {float(r["MaxNs"]) / 1e3:8.2f}')
PY
tail -4 gpurun_out/r03/k1_split_trace.log
