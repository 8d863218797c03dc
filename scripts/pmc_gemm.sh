#!/bin/bash
# Counter passes over one GEMM of the micro-benchmark (GEMM_ONLY selects shape and product); one rocprofv3
# run per pass.  usage: scripts/pmc_gemm.sh [1600x400:fwd]
set -e
export TMPDIR=/tmp; R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/pmc; rm -rf $O; mkdir -p $O; cd /tmp
export GEMM_ONLY=${1:-1600x400:fwd}
run() { n=$1; shift; timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- python $R/scripts/gemm_bench.py bf16 > $O/$n.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_SCA
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_TD_BUSY_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_REQ_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for f in sorted(glob.glob(O + "/**/*_counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "gemm_kernel" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        print(f"{k:40s} per launch {v / n:16.0f}   (n={n})")
PY
