#!/bin/bash
# round 5, first GPU call: the new / changed tests, then the three bench lines on this box (baseline for the round's A/Bs)
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/r5_first; mkdir -p $O
timeout 1500 python -m pytest -q -x -m gpu tests/test_hip_multigpu.py tests/test_sharding.py \
  "tests/test_train.py::test_bf16_arm_lm_head_backward_uses_current_weights_when_rows_are_not_a_multiple_of_64" \
  "tests/test_train.py::test_checkpoint_save_then_finetune_entry_point" \
  "tests/test_hip_bf16.py::test_decoder_bf16_arm_with_a_top_level_the_bf16_activation_stream_cannot_take" \
  "tests/test_hip_bf16.py::test_decoder_bf16_pixels_within_stated_tolerance_and_tokens_stay_exact" \
  "tests/test_hip_fp8.py" "tests/test_hip_parity_scale.py::test_mixed_arm_end_to_end_against_oracle" > $O/tests.log 2>&1
echo "pytest rc=$?"; grep -E "^E  |passed|failed|error" $O/tests.log | cut -c1-300 | head -30
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench_views.json 2> $O/bench_views.err; echo "bench rc=$?"; cut -c1-200 $O/bench_views.json
timeout 600 python bench.py --workload train --steps 10 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err; echo "train rc=$?"; cut -c1-200 $O/bench_train.json
timeout 600 python bench.py --views 20 --steps 5 --warmup 2 --no-cpu-baseline --no-f32-arm > $O/bench_s20.json 2> $O/bench_s20.err; echo "s20 rc=$?"; cut -c1-200 $O/bench_s20.json
