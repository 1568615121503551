#!/bin/bash
# round 4, GPU call A: what the round-3 code costs at the reference's dropout 0.1 (bf16 arm falls back to the f32 attention etc.)
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4a
timeout 300 python bench.py --workload train --steps 10 --warmup 2 --dropout 0.0 > gpurun_out/r4a/train_d0.json 2> gpurun_out/r4a/train_d0.err; echo "d0 rc=$?"; cut -c1-300 gpurun_out/r4a/train_d0.json
timeout 300 python bench.py --workload train --steps 10 --warmup 2 --dropout 0.1 > gpurun_out/r4a/train_d01.json 2> gpurun_out/r4a/train_d01.err; echo "d01 rc=$?"; cut -c1-300 gpurun_out/r4a/train_d01.json
bash tools/prof_train.sh r4a_d01 --dropout 0.1 --precision bf16 > gpurun_out/r4a/prof_summary.txt 2>&1; echo "prof rc=$?"
head -30 gpurun_out/r4a/prof_summary.txt | cut -c1-200
