#!/bin/bash
# Run the GPU test-suite on the box; logs go to gpurun_out/ (merged back by gpurun).
# usage: scripts/gpu_tests.sh [pytest args...]
set -o pipefail
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -u -m pytest tests -m gpu -v --timeout 180 --timeout-method thread "$@" 2>&1 | tee gpurun_out/gpu_tests.log | grep -E "PASSED|FAILED|ERROR|passed|failed|error" | tail -150
