#!/bin/bash
# round 3, GPU call W: pre-activation saved as bf16, LayerNorm backward two rows in flight
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r3w
timeout 300 python -m pytest tests/test_hip_train_full.py tests/test_train.py -m gpu -v -s > gpurun_out/r3w/a.log 2>&1; echo "a rc=$?"; grep -E "^E  |passed|failed|worst|trajectory" gpurun_out/r3w/a.log | cut -c1-250 | head -20
timeout 300 python bench.py --workload train --steps 10 --warmup 2 > gpurun_out/r3w/train.json 2> gpurun_out/r3w/train.err; echo "train rc=$?"; cut -c1-240 gpurun_out/r3w/train.json
bash tools/prof_train.sh r3w_train --precision bf16 > gpurun_out/r3w/prof_train.txt 2>&1; head -32 gpurun_out/prof_r3w_train/summary.txt | cut -c1-170
