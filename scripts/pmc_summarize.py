#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; both in KiB).
On gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streaming reads at 64 bytes, so the
read side is doubled; WRITE_SIZE is exact for 16-byte streaming stores and for float atomics.
usage: pmc_summarize.py <dir with fetch/ and write/> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(\w+<[^(]*>|\w+)\(", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
        name = m.group(1) if m else r["Kernel_Name"]
        a = acc[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


def short_name(full):
    m = re.search(r"(\w+<[^(]*>|\w+)\(", full.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else full


def durations(d):
    """{kernel: average launch duration in us} from the kernel trace of the same counter pass (durations under a
    counter pass run a little long; they are there to date the figures, not to be quoted)."""
    fs = glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0, 0.0])
    if fs:
        for r in csv.DictReader(open(fs[0])):
            a = acc[short_name(r["Kernel_Name"])]
            a[0] += 1
            a[1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
    return {k: v[1] / v[0] for k, v in acc.items() if v[0]}


def main():
    d, out = sys.argv[1], sys.argv[2]
    fe, wr = per_kernel(d + "/fetch", "FETCH_SIZE"), per_kernel(d + "/write", "WRITE_SIZE")
    dur = durations(d + "/fetch")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        nf, sf = fe.get(k, [0, 0.0])
        nw, sw = wr.get(k, [0, 0.0])
        rd = 2.0 * 1024.0 * sf / max(nf, 1)
        wb = 1024.0 * sw / max(nw, 1)
        res[k] = dict(launches=nf, read_bytes_per_launch=round(rd), write_bytes_per_launch=round(wb), traffic_bytes_per_launch=round(rd + wb),
                      avg_launch_us=round(dur[k], 2) if k in dur else None)
    # bytes per optimizer step: everything the process moved, over the number of steps it ran (one Adam launch each;
    # the synthetic-data generation at start-up is included, a fraction of a percent)
    steps = max((v["launches"] for k, v in res.items() if k.startswith("adam")), default=0)
    total = sum(v["traffic_bytes_per_launch"] * v["launches"] for v in res.values())
    per_step = round(total / steps) if steps else None
    json.dump(dict(steps=steps, traffic_bytes_per_step=per_step, source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bench.py --no-sampler --no-vae --steps 3 --warmup 1",
                   correction="read = 2 x FETCH_SIZE x 1024 (gfx950 tallies 128-byte read requests at 64 bytes); write = WRITE_SIZE x 1024",
                   kernels=res), open(out, "w"), indent=1)
    print(f"steps {steps}, HBM traffic per step {per_step / 1e9 if per_step else 0:.2f} GB")
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{v['launches']:5d} x {v['traffic_bytes_per_launch'] / 1e6:9.1f} MB  (rd {v['read_bytes_per_launch'] / 1e6:8.1f} wr {v['write_bytes_per_launch'] / 1e6:8.1f})  {k}")


if __name__ == "__main__":
    main()
