#!/bin/bash
# round 4, GPU call F: attention after the visibility-mask change (tests, timing, stamps); MLP chunking experiment
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4f
timeout 600 python -m pytest tests/test_hip_bf16.py -m gpu -q -x -k "attention" > gpurun_out/r4f/a.log 2>&1; echo "tests rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4f/a.log | cut -c1-200 | head
for i in 1 2; do
for q in 1 0; do
VF_ATTN_Q32=$q python tools/microbench.py attnbf16_io16 attnbf16_train attnbf16_s20 2>&1 | grep attn
done
done
VF_HIP_LIB=$PWD/viewformer_amd/variants/libvf_stamps.so python tools/microbench.py attn_stamps > gpurun_out/r4f/stamps.txt 2>&1; grep -v amdgpu gpurun_out/r4f/stamps.txt | head -14 | cut -c1-200
python tools/bench_mlp_chunks.py 2>&1 | grep -v amdgpu | tee gpurun_out/r4f/mlp_chunks.txt
