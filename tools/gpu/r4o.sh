#!/bin/bash
# round 4, GPU call O: new tests (shared-GPU reproducibility, RCCL communicator of one rank)
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4o
timeout 900 python -m pytest tests/test_hip_repro.py tests/test_hip_multigpu.py -m gpu -q -x -k "repro or rccl_communicator" > gpurun_out/r4o/a.log 2>&1; echo "tests rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4o/a.log | cut -c1-300 | head -20; tail -30 gpurun_out/r4o/a.log | cut -c1-300 | grep -v "^$" | tail -15
