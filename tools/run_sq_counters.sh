# two rocprofv3 --pmc passes (8 SQ counters each, never combined with trace domains beyond --kernel-trace) + digest
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sq
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/sqA -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-graph --inflight 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/sqB -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-graph --inflight 1 > /dev/null 2>&1
python $R/tools/pmc_sq_summary.py /tmp/sqA /tmp/sqB > $R/gpurun_out/sq/sq_counters.txt 2>&1
head -30 $R/gpurun_out/sq/sq_counters.txt
