#!/bin/bash
# round 4, GPU call M: phase stagger of the two workgroups of a CU in the x3h convolution
for so in base r0 r1 r2; do
  VF_HIP_LIB=$PWD/viewformer_amd/variants/libvf_$so.so python tools/microbench.py convx3h x3h_stamps 2>&1 | grep -v amdgpu.ids | sed "s/^/[$so] /"
done
