#!/usr/bin/env python
"""In-process A/B of the inference step (bench.py's default workload: 128 scenes x 7 views) over BUILDS of the library (loaded side by side, every op of a
block of steps routed through one of them: viewformer_amd._lib.use), timed in alternating blocks on one box.
usage: python tools/ab_inprocess_views_libs.py viewformer_amd/libvf_hip.so viewformer_amd/variants/libvf_<name>.so ...   (AB_ROUNDS, AB_STEPS, AB_BATCH)"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from viewformer_amd import _lib  # noqa: E402
from viewformer_amd.evaluate import generate_batch_predictions  # noqa: E402
from viewformer_amd.weights import synthetic_scene_batch  # noqa: E402

libs = {os.path.basename(p): _lib.load_variant(p) for p in sys.argv[1:]}
vals = list(libs)
B = int(os.environ.get('AB_BATCH', 128))
rounds = int(os.environ.get('AB_ROUNDS', 6))
steps = int(os.environ.get('AB_STEPS', 8))
dev = torch.device('cuda:0')
vq, tr, _ = bench.build_models(dev, True, 'mixed')
frames, cams = synthetic_scene_batch(B, 7, 128, seed=0)
fr, cm = torch.from_numpy(frames).to(dev), torch.from_numpy(cams).to(dev)
for _ in range(3):
    out = generate_batch_predictions(tr, vq, fr, cm)
torch.cuda.synchronize()
ms = {v: [] for v in vals}
outs = {}
for r in range(rounds):
    for v in (vals if r % 2 == 0 else vals[::-1]):
        with _lib.use(libs[v]):
            generate_batch_predictions(tr, vq, fr, cm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(steps):
                out = generate_batch_predictions(tr, vq, fr, cm)
            e1.record()
            torch.cuda.synchronize()
            ms[v].append(e0.elapsed_time(e1) / steps)
            outs[v] = out
same = all(torch.equal(outs[vals[0]][k], outs[v][k]) for v in vals for k in ('generated_images', 'generated_cameras'))
print(json.dumps({'libraries': vals, 'batch': B, 'ms_per_step_median': {k: round(statistics.median(v), 3) for k, v in ms.items()},
                  'ms_per_step_all': {k: [round(x, 3) for x in v] for k, v in ms.items()}, 'outputs_identical': same}))
