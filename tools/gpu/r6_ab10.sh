#!/bin/bash
# full kernel lists (every kernel, by launch shape) of one inference and one training bench run; dQ ring A/B with the roles swapped
set -u
mkdir -p gpurun_out/r6_ab10
export PROF_ALL=1
bash tools/prof_bench.sh ab10_views --steps 3 --warmup 1 > gpurun_out/r6_ab10/views.log 2>&1; echo "views rc=$?"
bash tools/prof_bench.sh ab10_train --workload train --serial-wgrad --steps 3 --warmup 1 > gpurun_out/r6_ab10/train.log 2>&1; echo "train rc=$?"
timeout 600 python tools/ab_inprocess_train_attn.py viewformer_amd/libvf_hip.so viewformer_amd/variants/libvf_dq_ring3.so > gpurun_out/r6_ab10/dq_ring.json 2> gpurun_out/r6_ab10/dq_ring.err; echo "ab rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r6_ab10/dq_ring.json'))
for k in ('dropout_0.0','dropout_0.1'): print(k, d[k]['us_median'], d[k]['max_rel_diff_vs_first'])"
find gpurun_out -name "*.db" -delete
