#!/bin/bash
# round 3, GPU call F: colsum fix, train bench + trace, full GPU suite, bench line + kernel trace for profiles/
set -u
mkdir -p gpurun_out/r3f
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3f/pytest.log | cut -c1-300
timeout 300 python bench.py --workload train --steps 10 --warmup 2 > gpurun_out/r3f/train.json 2> gpurun_out/r3f/train.err; echo "train rc=$?"; cut -c1-260 gpurun_out/r3f/train.json
timeout 300 python bench.py --workload train --precision f32 --steps 6 --warmup 2 > gpurun_out/r3f/train_f32.json 2>> gpurun_out/r3f/train.err; cut -c1-260 gpurun_out/r3f/train_f32.json
bash tools/prof_train.sh r3f_train --precision bf16 > gpurun_out/r3f/prof_train.txt 2>&1; head -22 gpurun_out/prof_r3f_train/summary.txt | cut -c1-170
timeout 400 python bench.py --steps 10 --warmup 2 > gpurun_out/r3f/bench.json 2> gpurun_out/r3f/bench.err; echo "bench rc=$?"; cut -c1-200 gpurun_out/r3f/bench.json
