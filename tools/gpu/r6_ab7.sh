#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
E=gpurun_out/r6_ab7
mkdir -p $E
timeout 600 python tools/ab_inprocess_attn.py viewformer_amd/libvf_hip.so viewformer_amd/variants/libvf_attn_index_order.so > $E/ab_attn.jsonl 2> $E/ab_attn.err; echo "attn ab rc=$?"; cat $E/ab_attn.jsonl
timeout 600 python tools/ab_inprocess_train_attn.py viewformer_amd/libvf_hip.so viewformer_amd/variants/libvf_attn_index_order.so > $E/ab_train_attn.json 2> $E/ab_train_attn.err; echo "train attn ab rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6_ab7/ab_train_attn.json'))
for k in ('dropout_0.0','dropout_0.1'): print(k, d[k]['us_median'], d[k]['max_rel_diff_vs_first'])
PY
timeout 900 python -m pytest tests/test_hip_bf16.py tests/test_train.py tests/test_hip_ring_stress.py tests/test_hip_streams.py -q -k "attn or attention or flash or stream or ring" > $E/tests.log 2>&1; echo "tests rc=$?"; tail -2 $E/tests.log
for i in 1 2; do
  timeout 300 python bench.py --workload train --steps 20 --warmup 5 > $E/train_$i.json 2> $E/train_$i.err
  VF_HIP_LIB=$PWD/viewformer_amd/variants/libvf_attn_index_order.so timeout 300 python bench.py --workload train --steps 20 --warmup 5 > $E/train_idx_$i.json 2> $E/train_idx_$i.err
  timeout 300 python bench.py --views 20 --steps 5 --warmup 2 --no-cpu-baseline --no-f32-arm > $E/s20_$i.json 2> $E/s20_$i.err
  VF_HIP_LIB=$PWD/viewformer_amd/variants/libvf_attn_index_order.so timeout 300 python bench.py --views 20 --steps 5 --warmup 2 --no-cpu-baseline --no-f32-arm > $E/s20_idx_$i.json 2> $E/s20_idx_$i.err
  python - <<PY
import json
for n in ('train_$i','train_idx_$i'):
    d=json.load(open('$E/'+n+'.json')); print(n, d['value'], d['ms_per_step'])
for n in ('s20_$i','s20_idx_$i'):
    d=json.load(open('$E/'+n+'.json')); a=d['roofline']['attention']; print(n, d['value'], d['ms_per_step'], 'attention', a['avg_launch_us'], a['frac'])
PY
done
