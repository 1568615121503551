#!/usr/bin/env python
"""In-process A/B of the training step (bench.py --workload train: 10 scenes x 10 views, dropout 0.1, bf16 arm) over BUILDS of the library (loaded side by
side, every op of a block of steps routed through one of them: viewformer_amd._lib.use), timed in alternating blocks on one box.
usage: python tools/ab_inprocess_train_libs.py viewformer_amd/libvf_hip.so viewformer_amd/variants/libvf_<name>.so ..."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from viewformer_amd import _lib, geometry  # noqa: E402
from viewformer_amd.config import MIGTConfig  # noqa: E402
from viewformer_amd.migt import MIGT  # noqa: E402
from viewformer_amd.train import MIGTTrainer  # noqa: E402
from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch  # noqa: E402

libs = {os.path.basename(p): _lib.load_variant(p) for p in sys.argv[1:]}
vals = list(libs)
rounds = int(os.environ.get('AB_ROUNDS', 6))
steps = int(os.environ.get('AB_STEPS', 10))
dev = torch.device('cuda:0')
B, S = 10, 10
cfg = MIGTConfig(sequence_size=S, n_loss_skip=1, localization_weight='5', pose_multiplier=0.05, dropout=0.1, learning_rate=1e-4, weight_decay=0.05,
                 total_steps=40000, batch_size=B)
model = MIGT(cfg, precision='bf16').load_state_dict(make_migt_weights(cfg, seed=0)).to(dev)
tr = MIGTTrainer(model)
g = np.random.Generator(np.random.PCG64(0))
tokens = torch.from_numpy(g.integers(0, 1024, size=(B, S, 8, 8))).to(dev)
_, cams = synthetic_scene_batch(B, S, 8, seed=0)
poses = geometry.normalize_cameras(geometry.to_relative_cameras(torch.from_numpy(cams))[0]).to(dev)
for _ in range(3):
    tr.train_step(poses, tokens)
torch.cuda.synchronize()
ms = {v: [] for v in vals}
for r in range(rounds):
    order = vals if r % 2 == 0 else vals[::-1]
    for v in order:
        with _lib.use(libs[v]):
            tr.train_step(poses, tokens)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(steps):
                met = tr.train_step(poses, tokens)
            e1.record()
            torch.cuda.synchronize()
            ms[v].append(e0.elapsed_time(e1) / steps)
print(json.dumps({'libraries': vals, 'ms_per_step_median': {k: round(statistics.median(v), 3) for k, v in ms.items()},
                  'ms_per_step_all': {k: [round(x, 3) for x in v] for k, v in ms.items()}, 'loss_finite': bool(torch.isfinite(met['loss']))}))
