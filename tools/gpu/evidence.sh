#!/bin/bash
# the evidence run of a round (one gpurun call) — the whole -m gpu suite, smoke, the bench lines (driver's flags), kernel traces of the default
# line and of the training line (weight gradients serialised: durations not stretched by a concurrent kernel), the bare 2-rank launch, the
# multi-GPU pre-flight, and the HBM-traffic counter passes (FETCH_SIZE / WRITE_SIZE per launch shape) that bench.py's roofline.traffic quotes
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/evidence
E=gpurun_out/evidence
rm -f gpurun_out/parity_report.jsonl
timeout 2400 python -m pytest tests -m gpu -q > $E/gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|error" $E/gpu.log | cut -c1-250 | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $E/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $E/smoke.log | cut -c1-200
bash tools/prof_bench_pmc.sh evidence_views > $E/pmc_views.log 2>&1; echo "pmc views rc=$?"
bash tools/prof_train_pmc.sh evidence_train > $E/pmc_train.log 2>&1; echo "pmc train rc=$?"
python tools/make_pmc_json.py views=gpurun_out/pmc_evidence_views train=gpurun_out/pmc_evidence_train > $E/pmc_traffic.json 2> $E/pmc_traffic.err; echo "pmc json rc=$?"; wc -c $E/pmc_traffic.json
python -c "import json,sys; json.load(open(sys.argv[1]))" $E/pmc_traffic.json && cp $E/pmc_traffic.json profiles/r6_pmc_traffic.json   # the bench lines below quote THIS run's counters (per launch shape)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $E/bench_views.json 2> $E/bench_views.err; echo "bench rc=$?"; cut -c1-300 $E/bench_views.json
timeout 600 python bench.py --workload train --steps 20 --warmup 5 > $E/bench_train.json 2> $E/bench_train.err; echo "train rc=$?"; cut -c1-300 $E/bench_train.json
timeout 600 python bench.py --workload allimg --steps 5 --warmup 2 > $E/bench_allimg.json 2> $E/bench_allimg.err; echo "allimg rc=$?"; cut -c1-300 $E/bench_allimg.json
timeout 600 python bench.py --views 20 --steps 5 --warmup 2 --no-cpu-baseline --no-f32-arm > $E/bench_s20.json 2> $E/bench_s20.err; echo "s20 rc=$?"; cut -c1-300 $E/bench_s20.json
timeout 600 python bench.py --batch-sweep 1,8,64,256,1024 --steps 3 --warmup 1 > $E/bench_sweep.json 2> $E/bench_sweep.err; echo "sweep rc=$?"; cut -c1-300 $E/bench_sweep.json
VF_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 16 --no-cpu-baseline --no-f32-arm > $E/bench_bare_gpus2_gloo.json 2> $E/bench_bare_gpus2_gloo.err; echo "bare --gpus 2 (gloo) rc=$?"; cut -c1-200 $E/bench_bare_gpus2_gloo.json; grep '"preflight"' $E/bench_bare_gpus2_gloo.err | cut -c1-400
timeout 120 python bench.py --gpus 2 --steps 1 --warmup 0 > $E/bench_bare_gpus2_refused.out 2> $E/bench_bare_gpus2_refused.err; echo "bare --gpus 2 without gloo rc=$? (must be non-zero on a 1-GPU box)"; tail -1 $E/bench_bare_gpus2_refused.err | cut -c1-250
timeout 300 python bench.py --gpus 1 --preflight > $E/preflight_1rank_rccl.json 2> $E/preflight_1rank_rccl.err; echo "preflight 1 rank (RCCL) rc=$?"; cut -c1-500 $E/preflight_1rank_rccl.json
VF_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --preflight > $E/preflight_2ranks_gloo.json 2> $E/preflight_2ranks_gloo.err; echo "preflight 2 ranks (gloo) rc=$?"; cut -c1-500 $E/preflight_2ranks_gloo.json
bash tools/prof_bench.sh evidence --steps 3 --warmup 1 > $E/prof.log 2>&1; echo "prof rc=$?"; tail -3 $E/prof.log | cut -c1-200
bash tools/prof_bench.sh evidence_train --workload train --serial-wgrad --steps 3 --warmup 1 > $E/prof_train.log 2>&1; echo "prof train rc=$?"; tail -3 $E/prof_train.log | cut -c1-200
cp gpurun_out/parity_report.jsonl $E/parity_report.jsonl 2>/dev/null
find gpurun_out -name "*.db" -delete
