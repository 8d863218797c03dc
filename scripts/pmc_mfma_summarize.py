#!/usr/bin/env python3
"""Per-kernel matrix-pipe occupancy from the two counter passes of scripts/pmc_mfma.sh.
SQ_VALU_MFMA_BUSY_CYCLES sums, over the chip's 1024 SIMDs, the cycles a SIMD's matrix pipe was busy;
GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs.  busy fraction = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024).
usage: pmc_mfma_summarize.py <dir with train/ and sample/> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d):
    f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: dict(launches=0, mfma=0.0, gui=0.0))
    seen = set()
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+<[^(]*>|\w+)\(", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
        name = m.group(1) if m else r["Kernel_Name"]
        a = acc[name]
        key = (r.get("Dispatch_Id"), name)
        if key not in seen:
            seen.add(key)
            a["launches"] += 1
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            a["mfma"] += float(r["Counter_Value"])
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            a["gui"] += float(r["Counter_Value"])
    out = {}
    for k, a in acc.items():
        if a["gui"] <= 0:
            continue
        out[k] = dict(launches=a["launches"], mfma_busy_cycles_per_launch=round(a["mfma"] / a["launches"]),
                      gui_active_per_launch=round(a["gui"] / a["launches"]),
                      mfma_busy_frac=round(a["mfma"] / (a["gui"] / 8.0 * 1024.0), 4))
    return out


def main():
    d, out = sys.argv[1], sys.argv[2]
    res = dict(source="rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python scripts/mfma_probe.py train|sample",
               definition="mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), per kernel over its launches; "
                          "counter passes serialise the kernels (no overlap between streams)",
               train_step=per_kernel(d + "/train"), sampler=per_kernel(d + "/sample"))
    json.dump(res, open(out, "w"), indent=1)
    for sec in ("train_step", "sampler"):
        print("==", sec)
        for k, v in sorted(res[sec].items(), key=lambda kv: -kv[1]["gui_active_per_launch"] * kv[1]["launches"])[:14]:
            print(f"{v['launches']:5d} x  busy {v['mfma_busy_frac']:6.3f}  {k[:80]}")


if __name__ == "__main__":
    main()
