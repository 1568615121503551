"""ORACLE (test infrastructure, NOT product code) — CPU restatement of one MIGT training step.

Only ``tests/`` may import this module.  PARITY UNPINNED (TensorFlow absent, see migt_oracle.py): the
losses restate viewformer/models/migt.py:416-448 and QuaternionPoseRepresentation.call :156-177 on top of the
multi-stream forward of migt_oracle.migt_forward; gradients come from torch autograd (fp64) over that
restatement, and the optimizer restates AdamWeightDecay / WarmUp / CosineDecay / Keras Adam
(viewformer/models/utils.py:310-564; tensorflow==2.4.1 Adam: m,v update then
var -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import migt_oracle as mg


def schedule_value(s, t):
    s = str(s).strip()
    if s.startswith('cosine('):
        a, b, n = [float(x) for x in s[len('cosine('):-1].split(',')]
        return b + (a - b) * 0.5 * (math.cos(min(1.0, t / n) * math.pi) + 1.0)      # schedules.py:199-201
    return float(s)


def losses(sd, cfg, poses, tokens, step=0, dtype=torch.float64):
    """forward + losses of MIGT.train_step (migt.py:464-476) -> (scalar loss, metrics); autograd-enabled"""
    out = mg.migt_forward(sd, cfg, tokens, poses, dtype=dtype, compute_losses=True, grad=True)
    B, S = tokens.shape[:2]
    ids = tokens.reshape(B, S, -1).long()
    skip = cfg.n_loss_skip
    logits = out['logits'].reshape(B, S, ids.shape[-1], -1)
    ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), ids.reshape(-1), reduction='none').reshape(B, S, -1)   # :423
    ce = ce[:, skip:].mean((1, 2))                                                      # :424-426
    loss = ce * cfg.image_generation_weight                                             # :428
    metrics = dict(ce_loss=ce.mean())
    if cfg.use_localization:                                                            # :430-448
        hidden = out['hidden_states'][-1]                                               # the LOC stream
        raw = mg.mlp(sd, 'pose_criterion.pose_classifier', hidden, dtype)               # :157
        y = poses.to(dtype).unsqueeze(-2) * torch.tensor([cfg.pose_multiplier] * 3 + [1.0] * 4, dtype=dtype)   # :167
        pos = ((y[..., :3] - raw[..., :3]) ** 2).mean(-1)[:, skip:].mean((1, 2))        # :170-175
        ori = ((y[..., 3:] - raw[..., 3:]) ** 2).mean(-1)[:, skip:].mean((1, 2))
        w = schedule_value(cfg.localization_weight, step)                               # :446
        loss = loss + (pos + ori) * w                                                   # :440,447
        metrics.update(pose_pos_loss=pos.mean(), pose_ori_loss=ori.mean(), localization_weight=w)
    total = loss.mean()                                                                 # :476 reduce_mean
    metrics['loss'] = total
    return total, metrics


def gradients(sd_np, cfg, poses, tokens, step=0):
    """fp64 autograd gradients of the restated loss w.r.t. every variable"""
    sd = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in sd_np.items()}
    total, metrics = losses(sd, cfg, torch.as_tensor(poses), torch.as_tensor(tokens), step)
    total.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}
    return grads, {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in metrics.items()}


def learning_rate(step, init_lr, total_steps, warmup_steps):
    """WarmUp over CosineDecay (models/utils.py:346-361,403-412)"""
    if warmup_steps and step < warmup_steps:
        return init_lr * (step / warmup_steps)
    decay_steps = max(total_steps - warmup_steps, 1)
    t = min(step - warmup_steps, decay_steps)
    return init_lr * 0.5 * (1.0 + math.cos(math.pi * t / decay_steps))


def adam_weight_decay_step(params, grads, m, v, step, cfg, warmup_steps=2000, b1=0.9, b2=0.999, eps=1e-8):
    """in-place fp64 restatement of AdamWeightDecay._resource_apply_dense (models/utils.py:507-537)"""
    lr = learning_rate(step, cfg.learning_rate, cfg.total_steps, warmup_steps)
    t = step + 1
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    for k in params:
        if cfg.weight_decay > 0 and 'bias' not in k:          # exclude_from_weight_decay matches only "bias" here
            params[k] -= lr * params[k] * cfg.weight_decay
        m[k] = b1 * m[k] + (1 - b1) * grads[k]
        v[k] = b2 * v[k] + (1 - b2) * grads[k] ** 2
        params[k] -= lr_t * m[k] / (np.sqrt(v[k]) + eps)
    return lr
