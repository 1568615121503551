#!/bin/bash
# x3h16 staging transform on packed fp32 instructions vs the scalar form: isolated launches (bit-identity + time), then the whole inference step on both libraries
set -u
mkdir -p gpurun_out/r6_ab11
timeout 900 python tools/ab_inprocess_conv.py --cases s1res,s1,s1res64,s1res256 viewformer_amd/libvf_hip.so viewformer_amd/variants/libvf_x3h16_scalar_xform.so > gpurun_out/r6_ab11/conv.jsonl 2> gpurun_out/r6_ab11/conv.err; echo "conv rc=$?"
cut -c1-600 gpurun_out/r6_ab11/conv.jsonl
timeout 900 python tools/ab_inprocess_conv.py --cases s1res,s1 viewformer_amd/variants/libvf_x3h16_scalar_xform.so viewformer_amd/libvf_hip.so > gpurun_out/r6_ab11/conv_swapped.jsonl 2>> gpurun_out/r6_ab11/conv.err; echo "conv swapped rc=$?"
cut -c1-600 gpurun_out/r6_ab11/conv_swapped.jsonl
timeout 600 python -m pytest tests/test_hip_x3h.py tests/test_hip_parity_scale.py -m gpu -x -q 2>&1 | tail -3
