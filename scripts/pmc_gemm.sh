set -e
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/pmc; mkdir -p $O; cd /tmp
export GEMM_ONLY=1600x400:fwd
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/tcc -- python $R/scripts/gemm_bench.py bf16 > $O/tcc.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/sq -- python $R/scripts/gemm_bench.py bf16 > $O/sq.log 2>&1
find $O -name "*.csv" | head; 
