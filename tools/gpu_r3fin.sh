#!/bin/bash
# round 3, GPU final call: full GPU suite + bench lines + traces of the round-3 final code
set -u
mkdir -p gpurun_out/r3fin
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r3fin/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3fin/pytest.log | cut -c1-300
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3fin/bench.json 2> gpurun_out/r3fin/bench.err; echo "bench rc=$?"; cut -c1-160 gpurun_out/r3fin/bench.json
bash tools/prof_bench.sh r3fin_views --steps 4 --warmup 1 > gpurun_out/r3fin/prof_views.txt 2>&1; head -20 gpurun_out/prof_r3fin_views/summary.txt | cut -c1-170
timeout 300 python bench.py --workload allimg --steps 3 --warmup 1 > gpurun_out/r3fin/allimg.json 2>> gpurun_out/r3fin/bench.err; cut -c1-200 gpurun_out/r3fin/allimg.json
timeout 300 python bench.py --workload allimg --attention bf16 --steps 3 --warmup 1 > gpurun_out/r3fin/allimg_bf16.json 2>> gpurun_out/r3fin/bench.err; cut -c1-200 gpurun_out/r3fin/allimg_bf16.json
timeout 300 python bench.py --views 20 --batch 12 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3fin/s20.json 2>> gpurun_out/r3fin/bench.err; cut -c1-200 gpurun_out/r3fin/s20.json

timeout 300 python bench.py --workload train --steps 20 --warmup 3 > gpurun_out/r3fin/train.json 2>> gpurun_out/r3fin/bench.err; cut -c1-260 gpurun_out/r3fin/train.json
bash tools/prof_train.sh r3fin_train --precision bf16 > gpurun_out/r3fin/prof_train.txt 2>&1
find gpurun_out/prof_r3fin_views gpurun_out/prof_r3fin_train -name "*.db" -delete
