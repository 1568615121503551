#!/bin/bash
set -u
export PYTHONUNBUFFERED=1
E=gpurun_out/r6_ab4
mkdir -p $E
timeout 900 python tools/ab_inprocess_conv.py --cases s2,s2_64,s2_32 viewformer_amd/libvf_hip.so viewformer_amd/libvf_hip.so:6=0 viewformer_amd/variants/libvf_s2tap2.so viewformer_amd/variants/libvf_s2tap6.so > $E/ab_conv_s2.jsonl 2> $E/ab_conv_s2.err; echo "s2 ab rc=$?"; cut -c1-900 $E/ab_conv_s2.jsonl; tail -3 $E/ab_conv_s2.err
