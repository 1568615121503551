#!/bin/bash
# round 3, GPU call H: resident attention kernel (tests, A/B bench), full-size training test
set -u
mkdir -p gpurun_out/r3h
timeout 1200 python -m pytest tests/test_hip_bf16.py tests/test_hip_train_full.py tests/test_hip_models.py -m gpu -q -x > gpurun_out/r3h/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3h/pytest.log | cut -c1-300
for res in 1 0 1 0; do
VF_ATTN_RES=$res timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-arm > gpurun_out/r3h/bench_res$res.json 2>> gpurun_out/r3h/bench.err; python - <<PY
import json
d=json.load(open('gpurun_out/r3h/bench_res$res.json'))
a=d['roofline']['attention']
print('VF_ATTN_RES=$res', d['value'], d['ms_per_step'], a['kernel'][:20], a['avg_launch_us'], a['frac'], a['hbm']['frac'])
PY
done
