#!/bin/bash
# box-speed indicator (K1 at ~72 k rows of 512-B lines: 22 us on some boxes, 35 us on others) next to the translation / gather probe on the SAME box
set -o pipefail
mkdir -p gpurun_out/r03
T=gpurun_out/r03/box_$(date +%H%M%S).txt
(ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 REPS=1 timeout -k 10 600 python tools/k1_insitu.py "" 2>/dev/null | grep -v "^# setup"
timeout -k 10 200 tools/tlb_thrash_probe 2>&1 | grep "dependent chain\|fresh list  \|three output"; rocm-smi --showclocks 2>/dev/null | grep -i "clk") > $T
cat $T
