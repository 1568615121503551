#!/bin/bash
# round 3, GPU call O: two-rank launch line over gloo (after the rank-0-only collective fix), full-size training test, evaluate-loop tests
set -u
mkdir -p gpurun_out/r3o
timeout 900 python -m pytest tests/test_hip_train_full.py tests/test_train.py -m gpu -q -x > gpurun_out/r3o/pytest1.log 2>&1; echo "pytest1 rc=$?"; grep -E "passed|failed" gpurun_out/r3o/pytest1.log | tail -2; grep -E "^E  |FAILED" gpurun_out/r3o/pytest1.log | cut -c1-300 | head -8
timeout 700 python -m pytest tests/test_hip_multigpu.py -m gpu -q -x > gpurun_out/r3o/pytest2.log 2>&1; echo "pytest2 rc=$?"; grep -E "passed|failed" gpurun_out/r3o/pytest2.log | tail -2; grep -E "^E  |FAILED" gpurun_out/r3o/pytest2.log | cut -c1-400 | head -8
