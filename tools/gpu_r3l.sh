#!/bin/bash
# round 3, GPU call L: final-state evidence — full GPU suite, bench line, kernel traces (views + train), allimg / s20 lines
set -u
mkdir -p gpurun_out/r3l
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r3l/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3l/pytest.log | cut -c1-300
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3l/bench.json 2> gpurun_out/r3l/bench.err; echo "bench rc=$?"; cut -c1-160 gpurun_out/r3l/bench.json
bash tools/prof_bench.sh r3l_views --steps 4 --warmup 1 > gpurun_out/r3l/prof_views.txt 2>&1; head -20 gpurun_out/prof_r3l_views/summary.txt | cut -c1-170
timeout 300 python bench.py --workload allimg --steps 3 --warmup 1 > gpurun_out/r3l/allimg.json 2>> gpurun_out/r3l/bench.err; cut -c1-200 gpurun_out/r3l/allimg.json
timeout 300 python bench.py --workload allimg --attention bf16 --steps 3 --warmup 1 > gpurun_out/r3l/allimg_bf16.json 2>> gpurun_out/r3l/bench.err; cut -c1-200 gpurun_out/r3l/allimg_bf16.json
timeout 300 python bench.py --views 20 --batch 12 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3l/s20.json 2>> gpurun_out/r3l/bench.err; cut -c1-200 gpurun_out/r3l/s20.json
find gpurun_out/prof_r3l_views -name "*.db" -delete
