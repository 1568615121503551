#!/bin/bash
# round 4, GPU call C: the whole -m gpu suite after the dropout + hygiene changes, then smoke and the default bench line
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4c
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r4c/gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4c/gpu.log | cut -c1-250 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4c/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r4c/smoke.log | cut -c1-200
timeout 600 python bench.py > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r4c/bench.json
