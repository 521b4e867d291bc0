#!/bin/bash
# One SQ counter pass + a kernel trace of any command; prints per-dispatch means of the interesting kernels.
#   bash tools/pmc_quick.sh <tag> <command...>
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- "$@" > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT/pmc_sq -o pmc -- "$@" > $OUT/pmc.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -o pmc -- "$@" > $OUT/pmc2.log 2>&1
cd $REPO
python tools/prof_summary.py $OUT 2>&1 | grep -v "^ *$" | cut -c1-160
