#!/bin/bash
# round 6, call 2: GEMM tail tiles + attention 17th k-step — parity tests, in-process A/Bs
set -u
export PYTHONUNBUFFERED=1
E=gpurun_out/r6_ab1
mkdir -p $E
timeout 1500 python -m pytest tests/test_hip_bf16.py tests/test_hip_ring_stress.py -x -q > $E/tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 $E/tests1.log | cut -c1-300
timeout 900 python -m pytest "tests/test_hip_multigpu.py::test_preflight_world1_rccl_and_world2_gloo" tests/test_hip_multirank.py "tests/test_hip_parity_scale.py::test_mixed_arm_end_to_end_against_oracle" -x -q > $E/tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 $E/tests2.log | cut -c1-300
timeout 600 python tools/ab_gemm_tail.py viewformer_amd/variants/libvf_g256_r5.so > $E/ab_gemm_tail.jsonl 2> $E/ab_gemm_tail.err; echo "gemm ab rc=$?"; cut -c1-420 $E/ab_gemm_tail.jsonl
timeout 600 python tools/ab_inprocess_attn.py viewformer_amd/libvf_hip.so viewformer_amd/variants/libvf_k17off.so viewformer_amd/variants/libvf_dotsum.so > $E/ab_attn.jsonl 2> $E/ab_attn.err; echo "attn ab rc=$?"; cat $E/ab_attn.jsonl
timeout 600 python tools/ab_inprocess_train_attn.py viewformer_amd/libvf_hip.so viewformer_amd/variants/libvf_atbfold0.so > $E/ab_train_attn.json 2> $E/ab_train_attn.err; echo "train attn ab rc=$?"; cat $E/ab_train_attn.json
AB_QSCALE=1.0 timeout 600 python tools/ab_inprocess_train_attn.py viewformer_amd/libvf_hip.so viewformer_amd/variants/libvf_atbfold0.so > $E/ab_train_attn_bigscores.json 2> $E/ab_train_attn_bigscores.err; echo "train attn ab (|s| ~ 25) rc=$?"; cat $E/ab_train_attn_bigscores.json
timeout 600 python bench.py --workload train --steps 20 --warmup 5 > $E/bench_train.json 2> $E/bench_train.err; echo "train rc=$?"; cut -c1-200 $E/bench_train.json
VF_GEMM_TAIL=0 timeout 600 python tools/ab_inprocess_train_attn.py viewformer_amd/libvf_hip.so viewformer_amd/variants/libvf_atbfold0.so > $E/ab_train_attn.json 2> $E/ab_train_attn.err; echo "train attn ab rc=$?"; cat $E/ab_train_attn.json
AB_QSCALE=1.0 timeout 600 python tools/ab_inprocess_train_attn.py viewformer_amd/libvf_hip.so viewformer_amd/variants/libvf_atbfold0.so > $E/ab_train_attn_bigscores.json 2> $E/ab_train_attn_bigscores.err; echo "train attn ab (|s| ~ 25) rc=$?"; cat $E/ab_train_attn_bigscores.json
timeout 600 python bench.py --workload train --steps 20 --warmup 5 > $E/bench_train_tail0.json 2> $E/bench_train_tail0.err; echo "train tail0 rc=$?"; cut -c1-200 $E/bench_train_tail0.json
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $E/bench_views.json 2> $E/bench_views.err; echo "bench rc=$?"; cut -c1-200 $E/bench_views.json
grep -h parity gpurun_out/parity_report.jsonl 2>/dev/null | tail -2
