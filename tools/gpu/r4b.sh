#!/bin/bash
# round 4, GPU call B: dropout inside the bf16 training arm — tests, then the step at dropout 0.1 / 0.0, then a kernel trace
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_train.py tests/test_abi.py -m gpu -q -x > gpurun_out/r4b/a.log 2>&1; echo "test_train rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4b/a.log | cut -c1-250 | head -20
timeout 900 python -m pytest tests/test_hip_train_full.py -m gpu -q -x -s > gpurun_out/r4b/b.log 2>&1; echo "train_full rc=$?"; grep -E "^E  |passed|failed|error|worst|full-size" gpurun_out/r4b/b.log | cut -c1-250 | head -30
timeout 300 python bench.py --workload train --steps 10 --warmup 2 > gpurun_out/r4b/train_d01.json 2> gpurun_out/r4b/train_d01.err; echo "d01 rc=$?"; cut -c1-300 gpurun_out/r4b/train_d01.json
timeout 300 python bench.py --workload train --steps 10 --warmup 2 --dropout 0.0 > gpurun_out/r4b/train_d0.json 2> gpurun_out/r4b/train_d0.err; echo "d0 rc=$?"; cut -c1-300 gpurun_out/r4b/train_d0.json
bash tools/prof_train.sh r4b_d01 --dropout 0.1 --precision bf16 > gpurun_out/r4b/prof_summary.txt 2>&1; echo "prof rc=$?"
head -34 gpurun_out/r4b/prof_summary.txt | cut -c1-200
