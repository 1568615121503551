#!/bin/bash
# time every viewformer_amd/variants/libvf_*.so on the given microbench entries (GPU box)
for so in viewformer_amd/variants/libvf_*.so; do
  n=$(basename $so .so); n=${n#libvf_}
  VF_HIP_LIB=$PWD/$so python tools/microbench.py "$@" 2>&1 | grep -v amdgpu | sed "s/^/[$n] /"
done
