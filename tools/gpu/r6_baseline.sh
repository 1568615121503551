#!/bin/bash
# round 6, first call: this session's box baseline of the round-5 code (bench lines), PMC traffic passes, the transient probe
set -u
export PYTHONUNBUFFERED=1
E=gpurun_out/r6_base
mkdir -p $E
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 > $E/bench_views.json 2> $E/bench_views.err; echo "bench rc=$?"; cut -c1-200 $E/bench_views.json
timeout 600 python bench.py --workload train --steps 20 --warmup 5 > $E/bench_train.json 2> $E/bench_train.err; echo "train rc=$?"; cut -c1-200 $E/bench_train.json
bash tools/prof_bench_pmc.sh r6_views > $E/pmc_views.log 2>&1; echo "pmc views rc=$?"
bash tools/prof_train_pmc.sh r6_train > $E/pmc_train.log 2>&1; echo "pmc train rc=$?"
bash tools/gpu/transient.sh 150
