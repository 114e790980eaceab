#!/bin/bash
set -o pipefail
bash tools/r03_part2b.sh
timeout -k 10 300 tools/tlb_thrash_probe > gpurun_out/r03/tlb_thrash_probe_d.txt 2>&1; grep "streaming copy\|dependent chain" gpurun_out/r03/tlb_thrash_probe_d.txt
