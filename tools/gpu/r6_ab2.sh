#!/bin/bash
# round 6, call 3: residual prefetch A/B (x3h16), full -m gpu suite on the current code, transient probe with RAS counters
set -u
export PYTHONUNBUFFERED=1
E=gpurun_out/r6_ab2
mkdir -p $E
timeout 900 python tools/ab_inprocess_conv.py --cases s1res,s1,s1res64,s1res256 viewformer_amd/libvf_hip.so viewformer_amd/variants/libvf_respf0.so > $E/ab_conv.jsonl 2> $E/ab_conv.err; echo "conv ab rc=$?"; cat $E/ab_conv.jsonl | cut -c1-600
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-arm > $E/bench_views.json 2> $E/bench_views.err; echo "bench rc=$?"; cut -c1-200 $E/bench_views.json
VF_HIP_LIB=$PWD/viewformer_amd/variants/libvf_respf0.so timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-arm > $E/bench_views_respf0.json 2> $E/bench_views_respf0.err; echo "bench respf0 rc=$?"; cut -c1-200 $E/bench_views_respf0.json
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-arm > $E/bench_views2.json 2> $E/bench_views2.err; echo "bench again rc=$?"; cut -c1-200 $E/bench_views2.json
timeout 2400 python -m pytest tests -m gpu -q -x > $E/gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|error" $E/gpu.log | cut -c1-250 | head -20
bash tools/gpu/transient.sh 90
