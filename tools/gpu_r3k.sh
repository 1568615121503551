#!/bin/bash
# round 3, GPU call K: software-pipelined ring attention (bit-identity + A/B), full-size training test
set -u
mkdir -p gpurun_out/r3k
timeout 1200 python -m pytest tests/test_hip_bf16.py tests/test_hip_train_full.py -m gpu -q -x > gpurun_out/r3k/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3k/pytest.log | cut -c1-300
for pipe in 1 0 1 0; do
VF_ATTN_PIPE=$pipe timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-arm > gpurun_out/r3k/bench_pipe$pipe.json 2>> gpurun_out/r3k/bench.err; python - <<PY
import json
d=json.load(open('gpurun_out/r3k/bench_pipe$pipe.json'))
a=d['roofline']['attention']
print('VF_ATTN_PIPE=$pipe', d['value'], d['ms_per_step'], a['avg_launch_us'], a['frac'], a['hbm']['frac'])
PY
done
VF_ATTN_PIPE=1 timeout 300 python bench.py --workload train --steps 10 --warmup 2 2>/dev/null | cut -c1-220
VF_ATTN_PIPE=0 timeout 300 python bench.py --workload train --steps 10 --warmup 2 2>/dev/null | cut -c1-220
