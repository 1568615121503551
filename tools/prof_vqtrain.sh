#!/bin/bash
# kernel-trace stats of the codebook training-step bench (GPU box, via gpurun): bash tools/prof_vqtrain.sh <tag> [bench args]
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/tools/bench_vqtrain.py --steps 2 --warmup 1 "$@" > $OUT/trace.log 2>&1
cd $R
python tools/summarize_prof.py $OUT
tail -1 $OUT/trace.log | cut -c1-500
find $OUT -name "*.db" -delete
