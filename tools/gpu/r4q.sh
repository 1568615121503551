#!/bin/bash
# round 4, GPU call Q: whole-step A/B of the 16x16x32 x3h convolution, then the new tests
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4q
for k in 1 0 1 0; do
VF_CONV_X3H_K32=$k python bench.py --no-cpu-baseline --no-f32-arm --steps 6 --warmup 2 > gpurun_out/r4q/bench_k32_$k.json 2>gpurun_out/r4q/bench_k32_$k.err; echo "k32=$k rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r4q/bench_k32_$k.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['avg_launch_ms'])
PY
done
timeout 900 python -m pytest tests/test_hip_x3h.py tests/test_hip_parity_scale.py -m gpu -q -x -k "both_mfma or token_flip" > gpurun_out/r4q/a.log 2>&1; echo "tests rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4q/a.log | cut -c1-300 | head -12
