#!/bin/bash
# round 3, GPU call B: training-step changes (tests, bench, kernel trace), new parity tests
set -u
mkdir -p gpurun_out/r3b
timeout 1500 python -m pytest tests/test_train.py tests/test_hip_train_full.py tests/test_hip_multirank.py tests/test_hip_vq_filter.py "tests/test_hip_parity_scale.py" -m gpu -x -q -s > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3b/pytest.log | cut -c1-400
timeout 300 python bench.py --workload train --steps 10 --warmup 2 > gpurun_out/r3b/train.json 2> gpurun_out/r3b/train.err; echo "train rc=$?"; cut -c1-300 gpurun_out/r3b/train.json; tail -3 gpurun_out/r3b/train.err
timeout 300 python bench.py --workload train --precision f32 --steps 6 --warmup 2 > gpurun_out/r3b/train_f32.json 2>> gpurun_out/r3b/train.err; echo "train f32 rc=$?"; cut -c1-300 gpurun_out/r3b/train_f32.json
bash tools/prof_train.sh r3b_train --precision bf16 > gpurun_out/r3b/prof_train.txt 2>&1; tail -40 gpurun_out/r3b/prof_train.txt | cut -c1-200
