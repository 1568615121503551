#!/bin/bash
# round 4, GPU call J: the two-wave bf16-activation decoder convolution (identity tests, microbench, bench A/B)
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4j
timeout 900 python -m pytest tests/test_hip_bf16.py -m gpu -q -s -k "conv3 or decoder or pipeline or evaluator" > gpurun_out/r4j/a.log 2>&1; echo "tests rc=$?"; grep -E "^E  |passed|failed|error|decoder|uint8" gpurun_out/r4j/a.log | cut -c1-250 | head -20
python tools/microbench.py convbf16_dec convbf16_io16 convbf16_io16_nopro convbf16_io16_64 convbf16_io16_256 2>&1 | tail -6
for a in 0 1 0 1; do
python bench.py --no-cpu-baseline --no-f32-arm --decoder-act16 $a --steps 6 --warmup 2 > gpurun_out/r4j/bench_act16_$a.json 2>gpurun_out/r4j/bench_act16_$a.err; echo "act16=$a rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r4j/bench_act16_$a.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], [ (t['shape'], t['ms'], t['tflops']) for t in d['roofline']['top_shapes_mode_M_Cin_Cout_batch']])
PY
done
