#!/bin/bash
# round 3, GPU call T: the three files of call R again, verbose + unbuffered, short timeouts (call R's pytest ran into its 900 s limit)
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r3t
timeout 300 python -m pytest tests/test_hip_train_full.py tests/test_train.py -m gpu -v > gpurun_out/r3t/a.log 2>&1; echo "a rc=$?"; grep -E "^E  |passed|failed" gpurun_out/r3t/a.log | cut -c1-250 | head; tail -2 gpurun_out/r3t/a.log | cut -c1-200
timeout 300 python -m pytest tests/test_hip_bf16.py -m gpu -v > gpurun_out/r3t/b.log 2>&1; echo "b rc=$?"; grep -E "^E  |passed|failed" gpurun_out/r3t/b.log | cut -c1-250 | head; tail -2 gpurun_out/r3t/b.log | cut -c1-200
