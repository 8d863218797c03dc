#!/bin/bash
# Counter passes over the weight-gradient launch of one recurrent layer (scripts/tn_one.py); one rocprofv3 run per pass.
# usage: scripts/pmc_tn.sh [lstm0|lstm1|gru0]
set -e
export TMPDIR=/tmp; R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/pmc_tn; rm -rf $O; mkdir -p $O; cd /tmp
W=${1:-lstm0}
run() { n=$1; shift; timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python $R/scripts/tn_one.py $W > $O/$n.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_ACTIVE_INST_ANY
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for f in sorted(glob.glob(O + "/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "mm16_tn" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        print(f"{k:40s} per launch {v / n:16.0f}   (n={n})")
for f in sorted(glob.glob(O + "/sq/**/*_kernel_trace.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "mm16_tn" in r["Kernel_Name"]]
    if rows:
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
        print("kernel duration ns (profiled):", sum(d) / len(d), "VGPR", rows[0].get("VGPR_Count"), "LDS", rows[0].get("LDS_Block_Size"), "grid", rows[0].get("Grid_Size_X"), "wg", rows[0].get("Workgroup_Size_X"))
PY
