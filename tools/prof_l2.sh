#!/bin/bash
# L2 hit / miss / fetch-size counters for microbench entries: bash tools/prof_l2.sh <out name> "<entries>" <kernel patterns...>
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$1; K="$2"; shift 2
mkdir -p $O
cd /tmp
MB="python $R/tools/microbench.py"
timeout 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $O/pmc3 -o p -- $MB $K > $O/pmc3.log 2>&1
timeout 90 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d $O/pmc4 -o p -- $MB $K > $O/pmc4.log 2>&1
cd $R
python tools/summarize_prof.py $O "$@" > $O/summary.txt 2>&1
find $R/gpurun_out -name "*.db" -delete
cat $O/summary.txt
