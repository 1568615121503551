#!/bin/bash
# round 3, GPU call C: 8-wave attention, deterministic embedding backward, full-size training tests, peaked-logit parity
set -u
mkdir -p gpurun_out/r3c
timeout 1800 python -m pytest tests/test_hip_bf16.py tests/test_train.py tests/test_hip_train_full.py tests/test_hip_multirank.py tests/test_hip_parity_scale.py -m gpu -q -s > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/r3c/pytest.log | tail -15 | cut -c1-300
grep -E "worst|full-size|peaked" gpurun_out/r3c/pytest.log | cut -c1-600
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-arm > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c/bench.json'))
print(d['value'], d['ms_per_step'], d['host_io'])
a=d['roofline']['attention']; print(a['kernel'], a['avg_launch_us'], a['frac'], a['hbm']['frac'])
PY
VF_ATTN_DMA8=0 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-f32-arm > gpurun_out/r3c/bench_dma4.json 2>> gpurun_out/r3c/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c/bench_dma4.json'))
print('dma4', d['value'], d['ms_per_step'])
a=d['roofline']['attention']; print(a['kernel'], a['avg_launch_us'], a['frac'], a['hbm']['frac'])
PY
timeout 300 python bench.py --workload train --steps 10 --warmup 2 > gpurun_out/r3c/train.json 2>> gpurun_out/r3c/bench.err; echo "train rc=$?"; cut -c1-260 gpurun_out/r3c/train.json
