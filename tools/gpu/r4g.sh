#!/bin/bash
# round 4, GPU call G: full suite after the ADVICE fixes + d_model 384 test; lookup stagger experiment; training bench
set -u
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4g
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r4g/gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|error" gpurun_out/r4g/gpu.log | cut -c1-250 | head -20
{
for rep in 1 2; do
python tools/microbench.py vqf vqf_big 2>&1 | grep vq_filtered | sed "s/^/[product] /"
for v in stag4k stag8k stag16k stag24k stag40k; do
VF_HIP_LIB=$PWD/viewformer_amd/variants/libvf_$v.so python tools/microbench.py vqf vqf_big 2>&1 | grep vq_filtered | sed "s/^/[$v] /"
done
done
} | cut -c1-200 | tee gpurun_out/r4g/vqf_stagger.txt
timeout 300 python bench.py --workload train --steps 10 --warmup 2 2>/dev/null | tee gpurun_out/r4g/train.json | cut -c1-250
