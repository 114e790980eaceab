#!/bin/bash
# GPU suite, then the rocprof / PMC passes and bench lines (tools/r03_final.sh) at HEAD
set -o pipefail
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/gputest_final.log 2>&1; echo "gputests rc=$?"; tail -4 gpurun_out/r03/gputest_final.log
bash tools/r03_final.sh
