#!/bin/bash
set -o pipefail
bash tools/r03_part2.sh
timeout -k 10 900 python tools/dist_config_probe.py > gpurun_out/r03/dist_config_probe.json 2> gpurun_out/r03/dist_config_probe.err; echo "dist_config_probe rc=$?"; tail -c 1500 gpurun_out/r03/dist_config_probe.json
