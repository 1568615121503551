#!/bin/bash
# round 4, GPU call E: 32-query wave tile of the LDS-DMA attention vs the 64-query form; stamps; PMC; tests
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4e
timeout 600 python -m pytest tests/test_hip_bf16.py tests/test_hip_parity_scale.py -m gpu -q -x -k "attention or activation_chain or s20" > gpurun_out/r4e/a.log 2>&1; echo "tests rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4e/a.log | cut -c1-200 | head
timeout 600 python -m pytest tests/test_train.py tests/test_hip_models.py tests/test_hip_train_full.py -m gpu -q -x -k "bf16 or fused or attention or dropout" > gpurun_out/r4e/b.log 2>&1; echo "tests2 rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4e/b.log | cut -c1-200 | head
for i in 1 2; do
for q in 1 0; do
VF_ATTN_Q32=$q python tools/microbench.py attnbf16_io16 attnbf16_train 2>&1 | grep attn
done
done
VF_HIP_LIB=$PWD/viewformer_amd/variants/libvf_stamps.so python tools/microbench.py attn_stamps > gpurun_out/r4e/stamps.txt 2>&1; cat gpurun_out/r4e/stamps.txt | cut -c1-200
bash tools/prof_kernel.sh r4e_attn "attnbf16_io16" attn_dma > gpurun_out/r4e/pmc.txt 2>&1; grep -A2 "attn_dma" gpurun_out/r4e/pmc.txt | cut -c1-330 | head -16
timeout 300 python bench.py --workload train --steps 10 --warmup 2 2>/dev/null | cut -c1-250
VF_ATTN_Q32=0 timeout 300 python bench.py --workload train --steps 10 --warmup 2 2>/dev/null | cut -c1-250
