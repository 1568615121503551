#!/bin/bash
# kernel-trace stats of one short bench run (GPU box, via gpurun): bash tools/prof_bench.sh <tag> [bench args]
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --no-cpu-baseline --no-f32-arm "$@" > $OUT/trace.log 2>&1
cd $R
python tools/summarize_prof.py $OUT
tail -2 $OUT/trace.log | cut -c1-600
