#!/bin/bash
# round 3, GPU call S: which of the new tests hangs (each under its own short timeout)
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r3s
timeout 120 python -m pytest tests/test_train.py -m gpu -q -k "layernorm_backward_bf16_copy" > gpurun_out/r3s/ln.log 2>&1; echo "ln rc=$?"; tail -3 gpurun_out/r3s/ln.log | cut -c1-200
timeout 120 python -m pytest tests/test_train.py -m gpu -q -k "gelu_dual" > gpurun_out/r3s/dual.log 2>&1; echo "dual rc=$?"; grep -E "^E  |passed|failed" gpurun_out/r3s/dual.log | cut -c1-250 | head
timeout 400 python -m pytest tests/test_hip_train_full.py -m gpu -q -s > gpurun_out/r3s/full.log 2>&1; echo "full rc=$?"; grep -E "^E  |passed|failed|worst|trajectory" gpurun_out/r3s/full.log | cut -c1-300 | head -20
