#!/bin/bash
# round 3, GPU call A: Winograd feed probe, full GPU test suite, bench line, batch sweep
set -u
mkdir -p gpurun_out/r3a
./tools/bin/wino_feed_probe > gpurun_out/r3a/wino_feed_probe.txt 2>&1
cat gpurun_out/r3a/wino_feed_probe.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3a/pytest.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r3a/bench.json
timeout 600 python bench.py --steps 3 --warmup 1 --batch-sweep 1,8,64,256,1024 > gpurun_out/r3a/sweep.json 2> gpurun_out/r3a/sweep.err; echo "sweep rc=$?"; cat gpurun_out/r3a/sweep.json
