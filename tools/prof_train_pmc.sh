#!/bin/bash
# HBM traffic counters of the training step's kernels (GPU box, via gpurun): bash tools/prof_train_pmc.sh <tag>
# two --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X guide, TCC counter budget), kernel-trace only — no other trace domain
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc3 -o p -- python $R/tools/bench_train.py --steps 1 --warmup 1 --precision bf16 > $O/pmc3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc4 -o p -- python $R/tools/bench_train.py --steps 1 --warmup 1 --precision bf16 > $O/pmc4.log 2>&1
cd $R
python tools/summarize_prof.py $O gemm_bf16_g256 gemm_tn_bf16 layernorm_bwd attn_ sum_slabs adamw > $O/summary.txt 2>&1
find $O -name "*.db" -delete
head -60 $O/summary.txt | cut -c1-200
