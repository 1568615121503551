#!/usr/bin/env python
"""In-process A/B of the inference step (bench.py's default workload: 128 scenes x 7 views) over a boolean module switch, timed in
alternating blocks on one box (box-to-box spread is +-3 %, the effects hunted here are < 1 %).
usage: python tools/ab_inprocess_views.py viewformer_amd.evaluate:CAMERA_SIDE_STREAM [batch] [rounds] [steps per block]"""
import importlib
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from viewformer_amd.evaluate import generate_batch_predictions  # noqa: E402
from viewformer_amd.weights import synthetic_scene_batch  # noqa: E402

modname, attr = sys.argv[1].split(':')
mod = importlib.import_module(modname)
while '.' in attr:                                    # module:Class.attr
    head, attr = attr.split('.', 1)
    mod = getattr(mod, head)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 6
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device('cuda:0')
vq, tr, _ = bench.build_models(dev, True, 'mixed')
frames, cams = synthetic_scene_batch(B, 7, 128, seed=0)
fr, cm = torch.from_numpy(frames).to(dev), torch.from_numpy(cams).to(dev)
for _ in range(3):
    out = generate_batch_predictions(tr, vq, fr, cm)
torch.cuda.synchronize()
ms = {False: [], True: []}
outs = {}
for r in range(rounds):
    for val in ((False, True) if r % 2 == 0 else (True, False)):
        setattr(mod, attr, val)
        generate_batch_predictions(tr, vq, fr, cm)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            out = generate_batch_predictions(tr, vq, fr, cm)
        e1.record()
        torch.cuda.synchronize()
        ms[val].append(e0.elapsed_time(e1) / steps)
        outs[val] = out
same = all(torch.equal(outs[False][k], outs[True][k]) for k in ('generated_images', 'generated_cameras'))
print(json.dumps({'switch': sys.argv[1], 'batch': B, 'ms_per_step_median': {str(k): round(statistics.median(v), 3) for k, v in ms.items()},
                  'ms_per_step_all': {str(k): [round(x, 3) for x in v] for k, v in ms.items()}, 'outputs_identical': same}))
