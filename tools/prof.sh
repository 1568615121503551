#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/prof.sh <tag> <microbench names...>
# kernel-trace stats in one run, PMC counters in separate runs (never combined with trace domains other than kernel-trace).
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $OLDPWD/tools/microbench.py "$@" > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc1 -o p -- python $OLDPWD/tools/microbench.py "$@" > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $OUT/pmc2 -o p -- python $OLDPWD/tools/microbench.py "$@" > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $OUT/pmc3 -o p -- python $OLDPWD/tools/microbench.py "$@" > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d $OUT/pmc4 -o p -- python $OLDPWD/tools/microbench.py "$@" > $OUT/pmc4.log 2>&1
cd $OLDPWD
find $OUT -name "*.csv" | head -30
python tools/summarize_prof.py $OUT
