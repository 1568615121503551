#!/bin/bash
# round 3, GPU call N: new evaluate-loop tests, two-rank launch line over gloo, full-size training test
set -u
mkdir -p gpurun_out/r3n
timeout 1500 python -m pytest tests/test_hip_evaluate_loop.py tests/test_hip_multigpu.py tests/test_hip_train_full.py -m gpu -q > gpurun_out/r3n/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r3n/pytest.log | tail -2; grep -E "^E  |FAILED|Error" gpurun_out/r3n/pytest.log | cut -c1-300 | head -20
