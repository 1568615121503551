#!/bin/bash
# round 3, GPU call E: bf16 training attention (tests, bench, trace), fixed full-size + peaked-logit tests
set -u
mkdir -p gpurun_out/r3e
timeout 1800 python -m pytest tests/test_train.py tests/test_hip_train_full.py tests/test_hip_multirank.py -m gpu -q -s > gpurun_out/r3e/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r3e/pytest.log | tail -3
grep -E "^E  |worst|trajectory|FAILED" gpurun_out/r3e/pytest.log | cut -c1-400 | head -40
grep mixed_arm_peaked gpurun_out/parity_report.jsonl | tail -1 | cut -c1-1500
timeout 300 python bench.py --workload train --steps 10 --warmup 2 > gpurun_out/r3e/train.json 2> gpurun_out/r3e/train.err; echo "train rc=$?"; cut -c1-260 gpurun_out/r3e/train.json; tail -2 gpurun_out/r3e/train.err
bash tools/prof_train.sh r3e_train --precision bf16 > gpurun_out/r3e/prof_train.txt 2>&1; head -24 gpurun_out/prof_r3e_train/summary.txt | cut -c1-180
