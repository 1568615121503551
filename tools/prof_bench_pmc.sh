#!/bin/bash
# HBM traffic counters of the inference step's kernels (GPU box, via gpurun): bash tools/prof_bench_pmc.sh <tag> [bench args]
# two --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X guide, TCC counter budget), kernel-trace only — no other trace domain
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc3 -o p -- python $R/bench.py --no-cpu-baseline --no-f32-arm --steps 1 --warmup 1 "$@" > $O/pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc4 -o p -- python $R/bench.py --no-cpu-baseline --no-f32-arm --steps 1 --warmup 1 "$@" > $O/pmc4.log 2>&1
cd $R
python tools/summarize_prof.py $O conv3_halo conv3_s2 gemm_bf16_g256 conv_in attn_dma attn_spatial gemm_x3h layernorm conv3_small vq_filter > $O/summary.txt 2>&1
find $O -name "*.db" -delete
head -70 $O/summary.txt | cut -c1-220
