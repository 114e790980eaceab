#!/bin/bash
# last call of the round: the overlap trace with the fixed summariser, and the two bench lines (bench.py's back-to-back check now rotates its output buffers too)
set -o pipefail
mkdir -p gpurun_out/r03
bash tools/dist_overlap_profile.sh > /dev/null 2>&1; cp gpurun_out/r02_dist_overlap.txt gpurun_out/r03/dist_overlap.txt; tail -4 gpurun_out/r03/dist_overlap.txt
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; echo "bench default rc=$?"
python bench.py --rows 111059956 --dim 128 --fanout 15,10,5 --cache-mb 16384 --epoch-steps 0 --no-fanout-leg --no-color-affinity-leg > gpurun_out/r03/bench_papers100m.json 2> gpurun_out/r03/bench_papers100m.err; echo "bench papers rc=$?"
python - <<'PY'
import json
for f in ("bench_default", "bench_papers100m"):
    d = json.load(open(f"gpurun_out/r03/{f}.json"))
    r = d["roofline"]; a = d.get("roofline_allhit") or {}
    print(f, d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], r["traffic"], a.get("frac"), a.get("avg_launch_us"), (a.get("back_to_back_check") or {}).get("kernel_time_bounds_us"), (a.get("mall_warm") or {}).get("frac"), (d.get("epoch") or {}).get("serial", {}).get("epoch_time_s_measured"), (d.get("epoch") or {}).get("prefetch", {}).get("epoch_time_s_measured"))
PY
