#!/bin/bash
# round 4, GPU call R: 512-channel layers on the 16x16x32 kernel (LDS fit), tests
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4r
timeout 900 python -m pytest tests/test_hip_x3h.py tests/test_hip_parity_scale.py -m gpu -q -x -k "both_mfma or token_flip or fp32_equivalent" > gpurun_out/r4r/a.log 2>&1; echo "tests rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4r/a.log | cut -c1-300 | head -12
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
sys.argv = ['x']
import tools.microbench as mb
from viewformer_amd import _lib
for k in (1, 0, 1, 0):
    _lib.select(_lib.SEL_CONV_X3H_K32, k)
    print('k32 =', k)
    mb.conv(224, 512, 16, x3h=True)
    mb.conv(56, 128, 128, x3h=True)
PY
