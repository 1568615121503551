#!/bin/bash
# round 3, GPU call J: TN weight-gradient GEMM, two-stage colsum restored
set -u
mkdir -p gpurun_out/r3j
timeout 1800 python -m pytest tests/test_train.py tests/test_hip_train_full.py -m gpu -q -s > gpurun_out/r3j/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r3j/pytest.log | tail -3
grep -E "^E  |full-size|FAILED" gpurun_out/r3j/pytest.log | cut -c1-300 | head -20
timeout 300 python bench.py --workload train --steps 10 --warmup 2 > gpurun_out/r3j/train.json 2> gpurun_out/r3j/train.err; echo "train rc=$?"; cut -c1-260 gpurun_out/r3j/train.json; tail -2 gpurun_out/r3j/train.err
bash tools/prof_train.sh r3j_train --precision bf16 > gpurun_out/r3j/prof_train.txt 2>&1; head -24 gpurun_out/prof_r3j_train/summary.txt | cut -c1-170
