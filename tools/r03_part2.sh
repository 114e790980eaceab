#!/bin/bash
# the probe files of profiles/ at HEAD (PART=2 of tools/profile_all.sh) + the round's K1 stage cuts on short lines
set -o pipefail
mkdir -p gpurun_out/r03
PART=2 ROUND=r03 bash tools/profile_all.sh > gpurun_out/r03/profile_part2.log 2>&1; echo "part2 rc=$?"; tail -20 gpurun_out/r03/profile_part2.log
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=12.75 timeout -k 10 600 python tools/k1_insitu.py --stages 2>/dev/null | grep -v "^# setup" > gpurun_out/profiles_out/r03_k1_stages_315k.body
ROWS=111059956 DIM=128 FANOUT=15,10,5 CACHE_MB=16384 DEG=6 timeout -k 10 600 python tools/k1_insitu.py --stages 2>/dev/null | grep -v "^# setup" > gpurun_out/profiles_out/r03_k1_stages_72k.body
timeout -k 10 600 python tools/k1_insitu.py --stages 2>/dev/null | grep -v "^# setup" > gpurun_out/profiles_out/r03_k1_stages_default.body
cat gpurun_out/profiles_out/r03_k1_stages_*.body
