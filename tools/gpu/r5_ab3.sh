#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_ab3; mkdir -p $O
V=$R/viewformer_amd/variants
cd /tmp
for n in default rg8; do
  L=""; [ $n != default ] && L=$V/libvf_$n.so
  VF_HIP_LIB=$L timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_$n -o p -- python $R/tools/microbench.py gemm_tf > $O/pmc_$n.log 2>&1
done
cd $R
python tools/summarize_prof.py $O gemm_bf16_g256 2>&1 | cut -c1-250
find $O -name "*.db" -delete
bash tools/gpu/r5_clock.sh
