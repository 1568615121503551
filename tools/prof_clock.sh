#!/bin/bash
# in-step vs isolated: duration and effective shader clock (GRBM_GUI_ACTIVE / wall) of the attention and the lookup kernels
# (the 15 % between the microbench figure and the timed step's figure — clock or access pattern?)
set -u
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/prof_clock; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmc_instep -o p -- python $R/bench.py --no-cpu-baseline --no-f32-arm --steps 2 --warmup 1 > $O/instep.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmc_isolated -o p -- python $R/tools/microbench.py attnbf16_io16 vqf convx3h > $O/isolated.log 2>&1
cd $R
python tools/summarize_prof.py $O attn_dma vq_filter conv3_halo_x3h16 2>&1 | cut -c1-200
find $O -name "*.db" -delete
