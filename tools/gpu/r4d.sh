#!/bin/bash
# round 4, GPU call D: folded-softmax attention (FOLD) vs the round-3 arithmetic (classic variant build), tests, PMC pass
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4d
timeout 600 python -m pytest tests/test_hip_bf16.py tests/test_hip_parity_scale.py -m gpu -q -x -s -k "attention or activation_chain or s20 or twin or fused" > gpurun_out/r4d/a.log 2>&1; echo "tests rc=$?"; grep -E "^E  |passed|failed|error|dma vs|12-layer" gpurun_out/r4d/a.log | cut -c1-200 | head -30
timeout 600 python -m pytest tests/test_train.py tests/test_hip_models.py -m gpu -q -x -k "bf16 or fused or attention" > gpurun_out/r4d/b.log 2>&1; echo "tests2 rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4d/b.log | cut -c1-200 | head
for i in 1 2; do
python tools/microbench.py attnbf16_io16 attnbf16_s20 2>&1 | grep attn | sed "s/^/[fold] /"
VF_HIP_LIB=$PWD/viewformer_amd/variants/libvf_classic.so python tools/microbench.py attnbf16_io16 attnbf16_s20 2>&1 | grep attn | sed "s/^/[classic] /"
done
bash tools/prof_kernel.sh r4d_attn "attnbf16_io16" attn_dma > gpurun_out/r4d/pmc.txt 2>&1; grep -A2 "attn_dma" gpurun_out/r4d/pmc.txt | cut -c1-330
timeout 300 python bench.py --workload train --steps 10 --warmup 2 2>/dev/null | cut -c1-250
