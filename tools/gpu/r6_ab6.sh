#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
E=gpurun_out/r6_ab6
mkdir -p $E
timeout 600 python tools/ab_gemm_tail.py viewformer_amd/variants/libvf_g256_r5.so > $E/ab_gemm_tail.jsonl 2> $E/ab_gemm_tail.err; echo "gemm ab rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r6_ab6/ab_gemm_tail.jsonl'):
    d=json.loads(l); print(d['case'][:46].ljust(46), d['tiles'], d['us_median'], d['same_bits_as_tail_off'])
PY
timeout 600 python -m pytest tests/test_hip_bf16.py -q -k "gemm" > $E/tests.log 2>&1; echo "tests rc=$?"; tail -2 $E/tests.log
for i in 1 2; do
  timeout 300 python bench.py --workload train --steps 20 --warmup 5 > $E/train_$i.json 2> $E/train_$i.err
  VF_HIP_LIB=$PWD/viewformer_amd/variants/libvf_g256_r5.so timeout 300 python bench.py --workload train --steps 20 --warmup 5 > $E/train_r5gemm_$i.json 2> $E/train_r5gemm_$i.err
  python - <<PY
import json
for n in ('train_$i','train_r5gemm_$i'):
    d=json.load(open('$E/'+n+'.json')); r=d['roofline']; print(n, d['value'], d['ms_per_step'], 'GEMM family', r['kernel_ms_per_step'], r['frac'])
PY
done
