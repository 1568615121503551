#!/bin/bash
# the evidence run of a round (one gpurun call) — the whole -m gpu suite, smoke, the three bench lines, kernel trace of the default line
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/evidence
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/evidence/gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/evidence/gpu.log | cut -c1-250 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/evidence/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/evidence/smoke.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/evidence/bench_views.json 2> gpurun_out/evidence/bench_views.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/evidence/bench_views.json
timeout 600 python bench.py --workload train > gpurun_out/evidence/bench_train.json 2> gpurun_out/evidence/bench_train.err; echo "train rc=$?"; cut -c1-300 gpurun_out/evidence/bench_train.json
timeout 600 python bench.py --workload allimg > gpurun_out/evidence/bench_allimg.json 2> gpurun_out/evidence/bench_allimg.err; echo "allimg rc=$?"; cut -c1-300 gpurun_out/evidence/bench_allimg.json
bash tools/prof_bench.sh evidence --steps 3 --warmup 1 > gpurun_out/evidence/prof.log 2>&1; echo "prof rc=$?"; tail -3 gpurun_out/evidence/prof.log | cut -c1-200
