#!/bin/bash
# round 3, GPU call Z2: TN weight-gradient kernel with XCD-contiguous (split-major) workgroup order
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r3z2
timeout 300 python -m pytest tests/test_train.py tests/test_hip_train_full.py -m gpu -q -x -k "tn_weight or full_size" > gpurun_out/r3z2/a.log 2>&1; echo "a rc=$?"; grep -E "^E  |passed|failed" gpurun_out/r3z2/a.log | cut -c1-250 | head
timeout 300 python bench.py --workload train --steps 10 --warmup 2 > gpurun_out/r3z2/train.json 2> gpurun_out/r3z2/train.err; echo "train rc=$?"; cut -c1-240 gpurun_out/r3z2/train.json
bash tools/prof_train_pmc.sh r3z2 > gpurun_out/r3z2/pmc.txt 2>&1; grep -A1 "gemm_tn_bf16" gpurun_out/pmc_r3z2/summary.txt | cut -c1-160
