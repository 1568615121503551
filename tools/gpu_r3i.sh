#!/bin/bash
# round 3, GPU call I: TN weight-gradient GEMM, two-stage colsum restored
set -u
mkdir -p gpurun_out/r3i
timeout 1800 python -m pytest tests/test_train.py tests/test_hip_train_full.py -m gpu -q -s > gpurun_out/r3i/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r3i/pytest.log | tail -3
grep -E "^E  |full-size|FAILED" gpurun_out/r3i/pytest.log | cut -c1-300 | head -20
timeout 300 python bench.py --workload train --steps 10 --warmup 2 > gpurun_out/r3i/train.json 2> gpurun_out/r3i/train.err; echo "train rc=$?"; cut -c1-260 gpurun_out/r3i/train.json; tail -2 gpurun_out/r3i/train.err
bash tools/prof_train.sh r3i_train --precision bf16 > gpurun_out/r3i/prof_train.txt 2>&1; head -24 gpurun_out/prof_r3i_train/summary.txt | cut -c1-170
