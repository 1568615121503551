#!/bin/bash
# Round-2 profiling session (GPU box, via gpurun): bash tools/prof_r2.sh
#  1. kernel trace of the default bench command                      -> gpurun_out/prof_r2_bench
#  2. PMC passes (each its own run, kernel-trace only) for the kernels the north star names, at bench shapes:
#     filtered codebook lookup, attention (bf16 / fp8 / x6 / f32), the dominant x3h convolution
#  3. TA / TCP (L1 path) counters of the x3h convolution, one counter group per run so that an unknown name only loses its own pass
#  4. power / clock samples (rocm-smi) while the x3h convolution runs in a loop
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof_r2
mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
MB="python $R/tools/microbench.py"
K="vqf vq_bench attnbf16 attnfp8 attnx6 attn convx3h"
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $MB $K > $O/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc1 -o p -- $MB $K > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $O/pmc2 -o p -- $MB $K > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $O/pmc3 -o p -- $MB $K > $O/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d $O/pmc4 -o p -- $MB $K > $O/pmc4.log 2>&1
i=5
for grp in "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" "TD_TD_BUSY_sum TCP_GATE_EN1_sum"; do
  rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$i -o p -- $MB convx3h convx6 > $O/pmc$i.log 2>&1
  i=$((i+1))
done
cd $R
python tools/summarize_prof.py $O vq_filter vq_argmin attn_ conv3_halo > $O/summary.txt 2>&1
# power / clocks under the dominant kernel
( for j in $(seq 1 14); do rocm-smi --showpower --showclocks --json 2>/dev/null | head -c 2000; echo; sleep 0.5; done ) > $O/smi_conv.txt &
python - <<PY > $O/power_loop.log 2>&1
import sys, time; sys.path.insert(0, '$R'); sys.argv=['x']
import torch, tools.microbench as mb
t0=time.time()
while time.time()-t0 < 6: mb.conv(x3h=True)
PY
wait
bash tools/prof_bench.sh r2_bench > $O/bench_trace.txt 2>&1
cp $R/gpurun_out/prof_r2_bench/summary.txt $O/bench_summary.txt 2>/dev/null
# keep what travels back small: the sqlite traces stay on the box
find $R/gpurun_out -name "*.db" -delete
tail -3 $O/trace.log
