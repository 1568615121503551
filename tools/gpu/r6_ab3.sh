#!/bin/bash
# round 6, call 4: stride-2 convolution with LDS-DMA staging (A/B, tap sweep, bit-identity tests), then the whole -m gpu suite
set -u
export PYTHONUNBUFFERED=1
E=gpurun_out/r6_ab3
mkdir -p $E
timeout 600 python -m pytest tests/test_hip_x3h.py -x -q > $E/tests_x3h.log 2>&1; echo "x3h tests rc=$?"; tail -3 $E/tests_x3h.log | cut -c1-300
cp viewformer_amd/libvf_hip.so /tmp/libvf_s2regs.so
cat > /tmp/ab_s2.py <<'PY'
import os, sys, ctypes
sys.path.insert(0, os.getcwd())
# the register-staged form = the product library with VF_SEL_CONV_S2_DMA off: a copy of the .so has its own switch state
PY
timeout 900 python - > $E/ab_conv_s2.jsonl 2> $E/ab_conv_s2.err <<'PY'
import os, sys, runpy
sys.path.insert(0, os.getcwd())
from viewformer_amd import _lib
regs = _lib.load_variant('/tmp/libvf_s2regs.so')
regs.vf_select(_lib.SEL_CONV_S2_DMA, 0)
sys.argv = ['ab', '--cases', 's2,s2_64,s2_32', 'viewformer_amd/libvf_hip.so', '/tmp/libvf_s2regs.so', 'viewformer_amd/variants/libvf_s2tap2.so', 'viewformer_amd/variants/libvf_s2tap6.so']
runpy.run_path('tools/ab_inprocess_conv.py', run_name='__main__')
PY
echo "s2 ab rc=$?"; cut -c1-700 $E/ab_conv_s2.jsonl; tail -3 $E/ab_conv_s2.err
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-arm > $E/bench_views.json 2> $E/bench_views.err; echo "bench rc=$?"; cut -c1-200 $E/bench_views.json
VF_CONV_S2_DMA=0 timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-arm > $E/bench_views_s2regs.json 2> $E/bench_views_s2regs.err; echo "bench s2regs rc=$?"; cut -c1-200 $E/bench_views_s2regs.json
timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-arm > $E/bench_views2.json 2> $E/bench_views2.err; echo "bench again rc=$?"; cut -c1-200 $E/bench_views2.json
VF_CONV_S2_DMA=0 timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-f32-arm > $E/bench_views_s2regs2.json 2> $E/bench_views_s2regs2.err; echo "bench s2regs again rc=$?"; cut -c1-200 $E/bench_views_s2regs2.json
timeout 2700 python -m pytest tests -m gpu -q > $E/gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^E  |passed|failed|error|FAILED" $E/gpu.log | cut -c1-250 | head -30
