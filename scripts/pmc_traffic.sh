#!/bin/bash
# HBM traffic per launch of every kernel of the training step, from the L2's memory-side counters:
# two rocprofv3 passes (FETCH_SIZE and WRITE_SIZE do not fit one pass) over the same short bench
# command; scripts/pmc_summarize.py turns them into profiles/<tag>_traffic.json.
# usage: scripts/pmc_traffic.sh <tag>
set -eo pipefail
TAG=${1:?tag}
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
O=$R/gpurun_out/pmc_$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- python "$R/bench.py" --no-cpu-baseline --no-sampler --no-vae --no-extras --steps 3 --warmup 1 > "$O/fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/write" -- python "$R/bench.py" --no-cpu-baseline --no-sampler --no-vae --no-extras --steps 3 --warmup 1 > "$O/write.log" 2>&1
python "$R/scripts/pmc_summarize.py" "$O" "$R/gpurun_out/${TAG}_traffic.json"
