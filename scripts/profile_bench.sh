#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command; the per-kernel statistics land in
# profiles/<tag>_kernel_stats.csv (committed) and the bench's JSON line in profiles/<tag>_bench.json.
# usage: scripts/profile_bench.sh <tag> [bench args...]
set -eo pipefail
TAG=${1:?tag}; shift || true
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
rm -rf "$O"; mkdir -p "$O" "$R/profiles" "$R/gpurun_out"
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python "$R/bench.py" --no-cpu-baseline "$@" > "$O/bench.log" 2>&1
grep '^{' "$O/bench.log" | tail -1 > "$R/gpurun_out/${TAG}_bench.json"
S=$(find "$O" -name '*_kernel_stats.csv' | head -1)
cp "$S" "$R/gpurun_out/${TAG}_kernel_stats.csv"
head -12 "$S" | cut -c1-200
