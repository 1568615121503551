#!/bin/bash
# round 4, GPU call K: decoder_act16 by starting resolution (tolerance + bench A/B)
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4k
timeout 900 python -m pytest tests/test_hip_bf16.py -m gpu -q -s -k "conv3 or decoder or pipeline or evaluator" > gpurun_out/r4k/a.log 2>&1; echo "tests rc=$?"; grep -E "^E  |passed|failed|error|decoder|uint8" gpurun_out/r4k/a.log | cut -c1-250 | head -20
for a in 0 128 32 0 128 32; do
python bench.py --no-cpu-baseline --no-f32-arm --decoder-act16 $a --steps 6 --warmup 2 > gpurun_out/r4k/bench_act16_$a.json 2>gpurun_out/r4k/bench_act16_$a.err; echo "act16=$a rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r4k/bench_act16_$a.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
PY
done
