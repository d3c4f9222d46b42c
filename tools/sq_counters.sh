#!/bin/bash
# Two rocprofv3 --pmc passes (SQ counters; never combined with trace domains other than --kernel-trace) over a command, digest by
# tools/pmc_sq_summary.py.   usage: tools/sq_counters.sh <tag> <command ...>     -> gpurun_out/<tag>_sq_counters.txt
tag=$1; shift
export TMPDIR=/tmp
root=$PWD
out=$PWD/gpurun_out
mkdir -p $out
da=$(mktemp -d /tmp/w2c_sqa_XXXX); db=$(mktemp -d /tmp/w2c_sqb_XXXX)
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $da -- "$@" > /dev/null 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $db -- "$@" > /dev/null 2>&1 )
python $root/tools/pmc_sq_summary.py $da $db $da > $out/${tag}_sq_counters.txt
rm -rf $da $db
