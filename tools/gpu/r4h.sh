#!/bin/bash
# round 4, GPU call H: weight-gradient GEMMs on a second stream; d_model 384 test; training tests
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4h
timeout 900 python -m pytest tests/test_train.py tests/test_hip_train_full.py tests/test_hip_multirank.py -m gpu -q -x -s > gpurun_out/r4h/a.log 2>&1; echo "tests rc=$?"; grep -E "^E  |passed|failed|error|d_model 384" gpurun_out/r4h/a.log | cut -c1-250 | head
for i in 1 2 3; do
python tools/bench_train.py --precision bf16 --steps 20 --warmup 3 2>/dev/null | cut -c1-300
python tools/bench_train.py --precision bf16 --steps 20 --warmup 3 --serial-wgrad 2>/dev/null | cut -c1-300
done
