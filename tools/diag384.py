"""bf16 training arm vs fp32-equivalent arm at toy widths (d_model 384 / 256), with and without dropout, bf16 or exact-f32 attention inside the bf16 arm: which tensors carry the largest relative gradient difference (diagnostic behind tests/test_train.py::test_bf16_training_arm_at_widths...)"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewformer_amd.config import MIGTConfig
from viewformer_amd.migt import MIGT
from viewformer_amd.train import MIGTTrainer
from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
from oracle import migt_oracle as mg
dev = torch.device('cuda:0')
for dm, nh in ((384, 6), (256, 4)):     # (std: pass 0.05 or 0.02 as argv[1])
  for dropout in (0.0, 0.1):
    cfg = MIGTConfig(n_layer=2, d_model=dm, n_head=nh, sequence_size=4, n_loss_skip=1, localization_weight='2', pose_multiplier=0.2,
                     dropout=dropout, learning_rate=1e-3, weight_decay=0.05, total_steps=50)
    sd = make_migt_weights(cfg, seed=4, std=float(sys.argv[1]) if len(sys.argv) > 1 else 0.02)
    g = np.random.Generator(np.random.PCG64(11))
    B, S = 2, 4
    tokens = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(B, S, 8, 8)))
    _, cams = synthetic_scene_batch(B, S, 8, 5)
    poses = mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(cams))[0])
    res = {}
    for arm, attn in (('f32', None), ('bf16', 'bf16'), ('bf16', 'f32')):
        tr = MIGTTrainer(MIGT(cfg, precision=arm).load_state_dict(sd).to(dev), warmup_steps=4)
        if attn: tr.attention_arith = attn
        tr.step_count, tr.dropout_seed = 3, 17
        m = tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
        res[(arm, attn)] = (tr.flat_g.clone(), float(m['loss']), tr)
    ref = res[('f32', None)][0]
    for key in (('bf16', 'bf16'), ('bf16', 'f32')):
        gk, loss, tr = res[key]
        errs = []
        for n in tr.names:
            a, b, _ = tr.slices[n]
            r = ref[a:b]
            if float(r.abs().max()) > 0:
                errs.append((((gk[a:b] - r).abs().max() / r.abs().max()).item(), n, float(r.abs().max())))
        errs.sort(reverse=True)
        print(dm, 'dropout', dropout, key, 'loss', loss, res[('f32', None)][1], 'worst:', [(round(e, 4), n, f'{mx:.2e}') for e, n, mx in errs[:4]])
