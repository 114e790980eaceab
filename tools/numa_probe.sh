#!/bin/bash
# K2 (zero-copy cold fill over PCIe) with the cold tier on the GPU's own NUMA node against the other socket's: the default bench
# workload twice, fetch-only, COALA_NUMA=auto (bind to the GPU's node before anything is pinned) and COALA_NUMA=far (the deliberate
# wrong placement).  TIER=host (hipHostMalloc: the runtime itself allocates on the GPU's node, whatever CPU asks) or TIER=shm (POSIX shm +
# hipHostRegister, the reference's kind: pages land where they are first touched).  -> gpurun_out/r03/numa_<tier>_{near,far}.json
set -o pipefail
mkdir -p gpurun_out/r03
X="--steps ${STEPS:-200} --epoch-steps 0 --no-fanout-leg --no-color-affinity-leg --no-cpu-baseline --no-allhit --cold-tier ${TIER:-host}"
T=${TIER:-host}
COALA_NUMA=auto python bench.py $X > gpurun_out/r03/numa_${T}_near.json 2> gpurun_out/r03/numa_${T}_near.err || exit 1
COALA_NUMA=far python bench.py $X > gpurun_out/r03/numa_${T}_far.json 2> gpurun_out/r03/numa_${T}_far.err || exit 1
python - $T <<'PY'
import json, sys
T = sys.argv[1]
for k in ("near", "far"):
    d = json.load(open(f"gpurun_out/r03/numa_{T}_{k}.json"))
    c, f = d["config"], d["roofline_cold_fill"]
    print(f"{T:4s} {k:4s}: gpu node {c['numa'][0].get('gpu_numa_node')}  bound node {c['numa'][0].get('bound_node')}  cold tier on node {c['cold_tier_numa_node']}  "
          f"K2 {f['achieved']:.2f} GB/s ({f['avg_launch_us']:.1f} us / launch)  step {d['ms_per_step']:.4f} ms  value {d['value']:.2f} GB/s")
PY
