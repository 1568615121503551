#!/bin/bash
# round 4, GPU call I: bf16 activations between the decoder's layers (kernel identity tests, model tolerance, bench A/B)
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4i
timeout 900 python -m pytest tests/test_hip_bf16.py -m gpu -q -x -s -k "conv3 or decoder or pipeline or evaluator" > gpurun_out/r4i/a.log 2>&1; echo "tests rc=$?"; grep -E "^E  |passed|failed|error|decoder|uint8" gpurun_out/r4i/a.log | cut -c1-250 | head -20
for a in 0 1 0 1; do
python bench.py --no-cpu-baseline --no-f32-arm --decoder-act16 $a --steps 6 --warmup 2 > gpurun_out/r4i/bench_act16_$a.json 2>gpurun_out/r4i/bench_act16_$a.err; echo "act16=$a rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r4i/bench_act16_$a.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v for k,v in d.items() if 'stage' in k or 'decod' in k})
PY
done
timeout 900 python -m pytest tests/test_hip_parity_scale.py tests/test_hip_models.py -m gpu -q -x > gpurun_out/r4i/b.log 2>&1; echo "tests2 rc=$?"; tail -3 gpurun_out/r4i/b.log | cut -c1-250
