#!/bin/bash
# HBM bytes and duration per launch of the weight-gradient launch of one recurrent layer (scripts/tn_one.py), two counter passes.
# usage: scripts/pmc_tn_traffic.sh [lstm0|lstm1|gru0]
set -e
export TMPDIR=/tmp; R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/pmc_tnt; rm -rf $O; mkdir -p $O; cd /tmp
W=${1:-lstm1}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- python $R/scripts/tn_one.py $W > $O/$c.log 2>&1
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
tot = {}
for f in sorted(glob.glob(O + "/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "mm16_tn" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        tot[k] = v / n
for f in sorted(glob.glob(O + "/FETCH_SIZE/**/*_kernel_trace.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "mm16_tn" in r["Kernel_Name"]]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    print("duration us:", round(sum(d) / len(d) / 1e3, 1), "grid", rows[0].get("Grid_Size_X"))
# gfx950: FETCH_SIZE counts half of a wide coalesced stream (units of 1 KiB... summarised as in scripts/pmc_summarize.py)
print({k: round(v) for k, v in tot.items()})
PY
