#!/bin/bash
# round 4, GPU call S: conflict-free LDS mapping of the 16x16x32 x3h convolution — tests, timing, PMC
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4s
timeout 900 python -m pytest tests/test_hip_x3h.py tests/test_hip_parity_scale.py -m gpu -q -x -k "both_mfma or token_flip or fp32_equivalent or partials" > gpurun_out/r4s/a.log 2>&1; echo "tests rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4s/a.log | cut -c1-300 | head -12
for k in 1 0 1 0; do VF_CONV_X3H_K32=$k python tools/microbench.py convx3h convx3h_64 convx3h_256 2>&1 | grep conv3x3 | sed "s/^/[k32=$k] /"; done
bash tools/prof_kernel.sh r4_x3h16 "convx3h" x3h16 2>&1 | grep -A1 "pmc2\|pmc3\|pmc4" | grep -v "^--" | cut -c1-250
