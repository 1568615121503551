#!/bin/bash
# round 3, GPU call Q: bf16 residual-stream gradient, GELU-backward epilogue in the 256-tile kernel
set -u
mkdir -p gpurun_out/r3q
timeout 900 python -m pytest tests/test_hip_train_full.py tests/test_train.py tests/test_hip_bf16.py -m gpu -q > gpurun_out/r3q/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r3q/pytest.log | tail -2; grep -E "^E  |FAILED" gpurun_out/r3q/pytest.log | cut -c1-300 | head -8
timeout 300 python bench.py --workload train --steps 10 --warmup 2 > gpurun_out/r3q/train.json 2> gpurun_out/r3q/train.err; echo "train rc=$?"; cut -c1-240 gpurun_out/r3q/train.json
bash tools/prof_train.sh r3q_train --precision bf16 > gpurun_out/r3q/prof_train.txt 2>&1; head -20 gpurun_out/prof_r3q_train/summary.txt | cut -c1-170
