#!/bin/bash
# round 3, final measurement call: rocprofv3 + PMC passes at HEAD (they also refresh profiles/pmc_probe_gather.json, which the bench
# lines below then quote as roofline.traffic), the full default bench line, the configs[3]-shape bench line
set -o pipefail
mkdir -p gpurun_out/r03
PART=1 ROUND=r03 bash tools/profile_all.sh > gpurun_out/r03/profile_part1.log 2>&1; echo "profiles rc=$?"
cp profiles/pmc_probe_gather.json gpurun_out/profiles_out/ 2>/dev/null
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err; echo "bench default rc=$?"
python bench.py --rows 111059956 --dim 128 --fanout 15,10,5 --cache-mb 16384 --epoch-steps 0 --no-fanout-leg --no-color-affinity-leg > gpurun_out/r03/bench_papers100m.json 2> gpurun_out/r03/bench_papers100m.err; echo "bench papers rc=$?"
python - <<'PY'
import json
for f in ("bench_default", "bench_papers100m"):
    d = json.load(open(f"gpurun_out/r03/{f}.json"))
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r["avg_launch_us"], r["traffic"], r["alg_bytes_per_launch"], (d.get("roofline_allhit") or {}).get("frac"), (d.get("epoch") or {}))
PY
