#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
V=$PWD/viewformer_amd/variants
mkdir -p gpurun_out/r5_ab4
python tools/ab_inprocess_attn.py viewformer_amd/libvf_hip.so $V/libvf_kg2.so $V/libvf_kg2mc.so $V/libvf_mc.so 2>/dev/null | tee gpurun_out/r5_ab4/attn_inprocess.jsonl
