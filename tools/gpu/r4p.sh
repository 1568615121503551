#!/bin/bash
# round 4, GPU call P: the 16x16x32 x3h convolution — kernel tests, token tests, A/B against the 32x32x16 kernel
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4p
timeout 900 python -m pytest tests/test_hip_x3h.py -m gpu -q -x > gpurun_out/r4p/a.log 2>&1; echo "x3h tests rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4p/a.log | cut -c1-300 | head -12
for k in 1 0 1 0; do VF_CONV_X3H_K32=$k python tools/microbench.py convx3h convx3h_64 convx3h_256 2>&1 | grep conv3x3 | sed "s/^/[k32=$k] /"; done
for k in 1 0; do VF_MB_ZERO=1 VF_CONV_X3H_K32=$k python tools/microbench.py convx3h 2>&1 | grep conv3x3 | sed "s/^/[zero k32=$k] /"; done
timeout 1200 python -m pytest tests/test_hip_models.py tests/test_hip_parity_scale.py -m gpu -q -x > gpurun_out/r4p/b.log 2>&1; echo "model tests rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4p/b.log | cut -c1-300 | head -12
