#!/bin/bash
# round 5: attention skew / permlane-swap A/B; g256 row-group order A/B with FETCH_SIZE; tests; bench
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_attn_skew; mkdir -p $O
V=$R/viewformer_amd/variants
for i in 1 2; do
  python tools/ab_attention_tn.py product$i >> $O/ab.jsonl 2>> $O/ab.err
  for n in noskew r4base skewdrop; do VF_HIP_LIB=$V/libvf_$n.so python tools/ab_attention_tn.py $n$i >> $O/ab.jsonl 2>> $O/ab.err; done
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/r5_attn_skew/ab.jsonl')]
tags=['product','noskew','r4base','skewdrop']
keys=[k for k in rows[0] if k.startswith('attn')]
for k in keys:
    base=[r for r in rows if r['tag'].startswith('r4base')][0][k]
    line=f'{k:22s}'
    for t in tags:
        rs=[r[k] for r in rows if r['tag'].startswith(t)]
        same=all(rs[0][x]==base[x] for x in base if x.startswith('digest'))
        us=[[r[x] for x in r if x.endswith('us')] for r in rs]
        line+=f' | {t}: same={same} us={us}'
    print(line)
PY
echo "== g256 tile order (gemm_tf at M = 65536)"
python tools/microbench.py gemm_tf 2>&1 | grep gemm_bf16 | sed "s/^/[default] /"
for n in rg4 rg8 rg16 rg8p0; do VF_HIP_LIB=$V/libvf_$n.so python tools/microbench.py gemm_tf 2>&1 | grep gemm_bf16 | sed "s/^/[$n] /"; done
python tools/microbench.py gemm_tf 2>&1 | grep gemm_bf16 | sed "s/^/[default] /"
cd /tmp
for n in default rg8 rg16; do
  L=""; [ $n != default ] && L=$V/libvf_$n.so
  VF_HIP_LIB=$L timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pmc_$n -o p -- python $R/tools/microbench.py gemm_tf > $O/pmc_$n.log 2>&1
done
cd $R
for n in default rg8 rg16; do echo "-- FETCH $n"; python tools/summarize_prof.py $O/pmc_$n gemm_bf16_g256 2>&1 | head -12 | cut -c1-250; done
find $O -name "*.db" -delete
timeout 1500 python -m pytest -q -m gpu tests/test_hip_bf16.py tests/test_hip_parity_scale.py tests/test_hip_streams.py tests/test_hip_models.py "tests/test_train.py" -k "attention or attn or twin or flash or mixed or migt or streams or bf16" > $O/tests.log 2>&1
echo "pytest rc=$?"; grep -E "^E  |passed|failed|error" $O/tests.log | cut -c1-300 | head -20
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-arm > $O/bench_views.json 2> $O/bench_views.err; cut -c1-160 $O/bench_views.json
VF_HIP_LIB=$V/libvf_r4base.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-arm > $O/bench_views_r4base.json 2> $O/bench_views_r4base.err; cut -c1-160 $O/bench_views_r4base.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-arm > $O/bench_views2.json 2> $O/bench_views2.err; cut -c1-160 $O/bench_views2.json
python bench.py --views 20 --steps 5 --warmup 2 --no-cpu-baseline --no-f32-arm > $O/bench_s20.json 2> $O/bench_s20.err; cut -c1-160 $O/bench_s20.json
