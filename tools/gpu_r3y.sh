#!/bin/bash
# round 3, GPU call Y: forward ring attention walks only the key tiles some wave of the workgroup sees
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r3y
timeout 400 python -m pytest tests/test_hip_bf16.py tests/test_train.py tests/test_hip_train_full.py tests/test_hip_fp8.py -m gpu -q -x > gpurun_out/r3y/a.log 2>&1; echo "a rc=$?"; grep -E "^E  |passed|failed" gpurun_out/r3y/a.log | cut -c1-250 | head
timeout 300 python bench.py --workload train --steps 10 --warmup 2 > gpurun_out/r3y/train.json 2> gpurun_out/r3y/train.err; echo "train rc=$?"; cut -c1-240 gpurun_out/r3y/train.json
bash tools/prof_train.sh r3y_train --precision bf16 > gpurun_out/r3y/prof_train.txt 2>&1; grep -E "attn_" gpurun_out/prof_r3y_train/summary.txt | cut -c1-170
