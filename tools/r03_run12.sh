#!/bin/bash
# same box: K1 on the configs[3] shape through bench.py (the manager allocates an output tensor per fetch) and through tools/k1_insitu.py (one output
# buffer for every fetch), same graph (avg degree 12)
set -o pipefail
mkdir -p gpurun_out/r03
python bench.py --rows 111059956 --dim 128 --fanout 15,10,5 --cache-mb 16384 --epoch-steps 0 --no-fanout-leg --no-color-affinity-leg --no-cpu-baseline --no-allhit > gpurun_out/r03/bench_papers100m_b.json 2> gpurun_out/r03/bench_papers100m_b.err; echo "bench papers rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/bench_papers100m_b.json"))
r = d["roofline"]
print("bench.py:", r["avg_launch_us"], "us", r["rows_per_launch"], "rows", r["frac"], "box copy", r.get("box_streaming_copy_gbs"))
PY
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12 SEGMENTS=1 REPS=2 timeout -k 10 900 python tools/k1_insitu.py "" > gpurun_out/r03/k1_insitu_deg12.txt 2> gpurun_out/r03/k1_insitu_deg12.err; echo "insitu rc=$?"; cat gpurun_out/r03/k1_insitu_deg12.txt
