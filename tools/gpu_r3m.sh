#!/bin/bash
# round 3, GPU call M: GELU backward fused into the mlp.c_proj dX GEMM epilogue
set -u
mkdir -p gpurun_out/r3m
timeout 1200 python -m pytest tests/test_hip_train_full.py tests/test_train.py tests/test_hip_bf16.py -m gpu -q > gpurun_out/r3m/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r3m/pytest.log | tail -2; grep -E "^E  |FAILED" gpurun_out/r3m/pytest.log | cut -c1-250 | head
timeout 300 python bench.py --workload train --steps 10 --warmup 2 > gpurun_out/r3m/train.json 2> gpurun_out/r3m/train.err; echo "train rc=$?"; cut -c1-240 gpurun_out/r3m/train.json
bash tools/prof_train.sh r3m_train --precision bf16 > gpurun_out/r3m/prof_train.txt 2>&1; head -16 gpurun_out/prof_r3m_train/summary.txt | cut -c1-170
