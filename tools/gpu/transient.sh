#!/bin/bash
# the GPU-sharing transient once more, with the platform's own error counters read before and after, under the default runtime
# settings and with HSA_ENABLE_SDMA=0 GPU_MAX_HW_QUEUES=1.  One gpurun call; output: gpurun_out/transient/*
set -u
export PYTHONUNBUFFERED=1
O=gpurun_out/transient
mkdir -p $O
SEC=${1:-150}
ras() { { date; rocm-smi --showrasinfo all 2>&1 | head -80; echo "--- dmesg tail"; dmesg 2>&1 | tail -30; } > $O/ras_$1.txt; }
ras before
timeout $((SEC + 300)) python tools/transient_probe.py 60 1 > $O/probe_1proc.jsonl 2> $O/probe_1proc.err; echo "1 process rc=$?"; tail -1 $O/probe_1proc.jsonl | cut -c1-300
timeout $((SEC + 300)) python tools/transient_probe.py $SEC 6 > $O/probe_default.jsonl 2> $O/probe_default.err; echo "default rc=$?"; tail -1 $O/probe_default.jsonl | cut -c1-400
ras after_default
HSA_ENABLE_SDMA=0 GPU_MAX_HW_QUEUES=1 timeout $((SEC + 300)) python tools/transient_probe.py $SEC 6 > $O/probe_nosdma_1queue.jsonl 2> $O/probe_nosdma_1queue.err; echo "no-sdma 1-queue rc=$?"; tail -1 $O/probe_nosdma_1queue.jsonl | cut -c1-400
ras after_nosdma_1queue
grep -c outlier_launch $O/probe_default.jsonl $O/probe_nosdma_1queue.jsonl
diff <(tail -n +2 $O/ras_before.txt) <(tail -n +2 $O/ras_after_nosdma_1queue.txt) > $O/ras_diff.txt; echo "ras diff lines: $(wc -l < $O/ras_diff.txt)"
