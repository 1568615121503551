#!/bin/bash
# PMC passes for one microbench entry set (GPU box, via gpurun): bash tools/prof_kernel.sh <out name> "<microbench entries>" <kernel name patterns...>
# -> gpurun_out/<out name>/summary.txt (per-kernel counters: wave cycles, waits, MFMA busy, LDS conflicts, ...)
set -u
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$1; K="$2"; shift 2
mkdir -p $O
cd /tmp
MB="python $R/tools/microbench.py"
timeout 60 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- $MB $K > $O/trace.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc1 -o p -- $MB $K > $O/pmc1.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $O/pmc2 -o p -- $MB $K > $O/pmc2.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -d $O/pmc3 -o p -- $MB $K > $O/pmc3.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum -d $O/pmc4 -o p -- $MB $K > $O/pmc4.log 2>&1
cd $R
python tools/summarize_prof.py $O "$@" > $O/summary.txt 2>&1
find $R/gpurun_out -name "*.db" -delete
cat $O/summary.txt
