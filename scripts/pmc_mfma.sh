#!/bin/bash
# Matrix-pipe occupancy per kernel for steady-state training steps and for sampler calls at the bench shape:
# one rocprofv3 counter pass each (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) over scripts/mfma_probe.py;
# scripts/pmc_mfma_summarize.py turns them into gpurun_out/<tag>_mfma.json (copy to profiles/).
# usage: scripts/pmc_mfma.sh <tag>
set -eo pipefail
TAG=${1:?tag}
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
O=$R/gpurun_out/mfma_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$O/train" -- python "$R/scripts/mfma_probe.py" train > "$O/train.log" 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$O/sample" -- python "$R/scripts/mfma_probe.py" sample > "$O/sample.log" 2>&1
python "$R/scripts/pmc_mfma_summarize.py" "$O" "$R/gpurun_out/${TAG}_mfma.json"
