#!/bin/bash
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r5_ab2; mkdir -p $O
V=$R/viewformer_amd/variants
python tools/ab_inprocess_attn.py $V/libvf_r4base.so $V/libvf_noskew.so viewformer_amd/libvf_hip.so 2>/dev/null | tee $O/attn_inprocess.jsonl
cd /tmp
for n in default rg8; do
  L=""; [ $n != default ] && L=$V/libvf_$n.so
  VF_HIP_LIB=$L timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_$n -o p -- python $R/tools/microbench.py gemm_tf > $O/pmc_$n.log 2>&1
done
cd $R
for n in default rg8; do echo "-- FETCH $n"; python tools/summarize_prof.py $O/pmc_$n gemm_bf16_g256 2>&1 | grep -v "^==" | head -12 | cut -c1-250; done
find $O -name "*.db" -delete
bash tools/gpu/r5_clock.sh
