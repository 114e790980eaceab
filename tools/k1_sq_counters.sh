#!/bin/bash
# Wave-scheduler and L2 counters of K1 in situ (rocprofv3 --pmc, separate passes): how long does a wave live, how much of that is it parked at a
# wait, how many waves are resident, what does the L2 say.  Usage on the GPU box:  [ROWS=.. DIM=.. FANOUT=.. CACHE_MB=.. DEG=..] bash tools/k1_sq_counters.sh <tag>
# HARNESS=tools/kernel_sq_probe.py: the same passes over the wide-grid cold fill (cold tier in HBM) and the un-permute kernel (SC below).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-papers100m}
export TMPDIR=/tmp REPS=1 GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
cd /tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CU_CYCLES"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
P4="GRBM_GUI_ACTIVE GRBM_COUNT"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i + 1))
  mkdir -p $R/gpurun_out/sq_${TAG}_$i
  timeout -k 10 400 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/sq_${TAG}_$i -- python3 $R/${HARNESS:-tools/k1_insitu.py} "" > $R/gpurun_out/sq_${TAG}_$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python3 - "$TAG" <<'PY'
import collections, csv, glob, os, sys
tag = sys.argv[1]
tot = collections.defaultdict(lambda: collections.Counter())
cnt = collections.defaultdict(lambda: collections.Counter())
for i in (1, 2, 3, 4):
    for f in glob.glob(f"gpurun_out/sq_{tag}_{i}/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        # steady state only: the last 200 launches of each kernel (the first 420 are warm-up minibatches with fewer hits)
        by = collections.defaultdict(list)
        for r in rows:
            k = "K1" if "probe_gather_kernel" in r["Kernel_Name"] else ("K2" if "miss_fill_kernel" in r["Kernel_Name"] else ("SC" if "scatter_rows_kernel" in r["Kernel_Name"] else None))
            if k:
                by[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in by.items():
            v = v[-200:]
            tot[k][c] = sum(v) / len(v)
for k in ("K1", "K2", "SC"):
    print(f"--- {k}: per launch, mean of the last 200 launches")
    for c, v in sorted(tot[k].items()):
        print(f"  {c:24s} {v:16.1f}")
    t = tot[k]
    if t.get("SQ_WAVES"):
        print(f"  wave lifetime            {4 * t['SQ_WAVE_CYCLES'] / t['SQ_WAVES']:12.0f} cycles (SQ_WAVE_CYCLES counts quad-cycles)")
        print(f"  parked at a wait         {100 * t['SQ_WAIT_ANY'] / t['SQ_WAVE_CYCLES']:8.1f} % of wave cycles; issue-stalled {100 * t['SQ_WAIT_INST_ANY'] / t['SQ_WAVE_CYCLES']:5.1f} %; issuing {100 * t['SQ_ACTIVE_INST_ANY'] / t['SQ_WAVE_CYCLES']:5.1f} %")
        print(f"  instructions per wave    VALU {t['SQ_INSTS_VALU'] / t['SQ_WAVES']:7.1f}  SALU {t['SQ_INSTS_SALU'] / t['SQ_WAVES']:7.1f}" + (f"  VMEM rd {t['SQ_INSTS_VMEM_RD'] / t['SQ_WAVES']:5.1f} wr {t['SQ_INSTS_VMEM_WR'] / t['SQ_WAVES']:5.1f} SMEM {t['SQ_INSTS_SMEM'] / t['SQ_WAVES']:5.1f}" if t.get("SQ_INSTS_VMEM_RD") else ""))
    if t.get("TCC_REQ_sum"):
        print(f"  L2 hit ratio             {100 * t['TCC_HIT_sum'] / max(t['TCC_HIT_sum'] + t['TCC_MISS_sum'], 1):8.1f} %  requests {t['TCC_REQ_sum']:.0f}")
PY
rm -rf $R/gpurun_out/sq_${TAG}_[1-4]   # the raw per-dispatch tables are hundreds of MB: only the summary travels back
