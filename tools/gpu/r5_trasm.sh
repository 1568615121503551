#!/bin/bash
# round 5: the transposing LDS reads as inline asm (no compiler-inserted vmcnt(0) in the DMA rings) against the intrinsic build, one box
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_trasm; mkdir -p $O
OLD=$PWD/viewformer_amd/variants/libvf_trintrin.so
for i in 1 2; do
  python tools/ab_attention_tn.py new$i >> $O/ab.jsonl 2>> $O/ab.err
  VF_HIP_LIB=$OLD python tools/ab_attention_tn.py old$i >> $O/ab.jsonl 2>> $O/ab.err
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/r5_trasm/ab.jsonl')]
new=[r for r in rows if r['tag'].startswith('new')]; old=[r for r in rows if r['tag'].startswith('old')]
for k in new[0]:
    if k in ('tag','lib'): continue
    a,b=new[0][k],old[0][k]
    same=all(a[x]==b[x] for x in a if x.startswith('digest'))
    tn=[ [r[k][x] for x in r[k] if x.endswith('us')] for r in new]; to=[ [r[k][x] for x in r[k] if x.endswith('us')] for r in old]
    print(f'{k:24s} bit-identical={same}  new us {tn}  old us {to}')
PY
timeout 1700 python -m pytest -q -m gpu tests/test_train.py tests/test_hip_bf16.py tests/test_hip_fp8.py tests/test_hip_parity_scale.py tests/test_hip_streams.py tests/test_hip_train_full.py > $O/tests.log 2>&1
echo "pytest rc=$?"; grep -E "^E  |passed|failed|error" $O/tests.log | cut -c1-300 | head -30
for i in 1 2; do
  python tools/bench_train.py --precision bf16 --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-300 | sed "s/^/[new] /"
  VF_HIP_LIB=$OLD python tools/bench_train.py --precision bf16 --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-300 | sed "s/^/[old] /"
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-arm > $O/bench_views_new.json 2> $O/bench_views_new.err; cut -c1-160 $O/bench_views_new.json
VF_HIP_LIB=$OLD python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-arm > $O/bench_views_old.json 2> $O/bench_views_old.err; cut -c1-160 $O/bench_views_old.json
python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train_new.json 2> $O/bench_train_new.err; cut -c1-200 $O/bench_train_new.json
python bench.py --workload allimg --steps 5 --warmup 2 > $O/bench_allimg_new.json 2> $O/bench_allimg_new.err; cut -c1-200 $O/bench_allimg_new.json
