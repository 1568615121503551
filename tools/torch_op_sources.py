#!/usr/bin/env python
"""Which host lines launch the small torch kernels of one evaluator batch (GPU box): torch.profiler with Python stacks over ONE
generate_batch_predictions call of the bench workload, aggregated by (aten op, innermost viewformer_amd frame).
usage: python tools/torch_op_sources.py [batch]"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from viewformer_amd.evaluate import generate_batch_predictions  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
vq, tr, _ = bench.build_models(dev, True, 'mixed')
from viewformer_amd.weights import synthetic_scene_batch  # noqa: E402
frames, cams = synthetic_scene_batch(B, 7, 128, seed=0)
fr, cm = torch.from_numpy(frames).to(dev), torch.from_numpy(cams).to(dev)
for _ in range(2):
    generate_batch_predictions(tr, vq, fr, cm)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    generate_batch_predictions(tr, vq, fr, cm)
    torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::'):
        continue
    if not ev.kernels:
        continue
    where = next((s for s in (ev.stack or []) if 'viewformer_amd' in s or 'bench.py' in s), '?')
    agg[(ev.name, where.split('viewformer_amd/')[-1][:110])] += len(ev.kernels)
tot = 0
for (name, where), n in agg.most_common(60):
    print(f'{n:4d}  {name:28s} {where}')
    tot += n
print('total device launches from aten ops:', sum(agg.values()))
