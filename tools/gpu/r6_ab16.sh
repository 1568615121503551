#!/bin/bash
# GN finalize on eight lanes per item: tests, profile of the inference step (all kernels by shape), bench line
python -m pytest tests/test_hip_ops.py tests/test_hip_x3h.py tests/test_hip_models.py -x -q -m gpu 2>&1 | tail -3
PROF_ALL=1 bash tools/prof_bench.sh ab16_views --steps 3 --warmup 1 > /dev/null 2>&1
grep -E "gn_finalize" gpurun_out/prof_ab16_views/summary*.txt | cut -c1-200
tail -1 gpurun_out/prof_ab16_views/trace.log | cut -c1-200
