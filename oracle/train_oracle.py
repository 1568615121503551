"""ORACLE (test infrastructure, NOT product code) — CPU restatement of one MIGT training step.

Only ``tests/`` may import this module.  PARITY UNPINNED (TensorFlow absent, see migt_oracle.py; anchored since round 5 to torch autograd
over Hugging Face GPT-2 on the same graph: loss terms to 1e-6, every gradient to 1e-8 — tests/golden/make_hf_gpt2_golden.py::train_graph): the
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


def random_pose_factors(cfg, B, seed, site=2):
    """rpm ** u_b with u_b = 2 * hash(seed, site, b) / 2^32 - 1: the build's counter-based stand-in for tf.random.uniform at
    migt.py:351 (train.py SITE_POSE_MULT = 2)"""
    u = dropout_hash(seed, site, np.arange(B, dtype=np.uint64)).astype(np.float64) / 2.0 ** 32 * 2.0 - 1.0
    return torch.from_numpy((float(cfg.random_pose_multiplier) ** u).astype(np.float32))


def losses(sd, cfg, poses, tokens, step=0, dtype=torch.float64, pose_factors=None):
    """forward + losses of MIGT.train_step (migt.py:464-476) -> (scalar loss, metrics); autograd-enabled.  ``pose_factors`` [B]: the
    per-scene random pose multiplier of this step (migt.py:350-354)"""
    out = mg.migt_forward(sd, cfg, tokens, poses, dtype=dtype, compute_losses=True, grad=True, pose_factors=pose_factors)
    B, S = tokens.shape[:2]
    ids = tokens.reshape(B, S, -1).long()
    skip = cfg.n_loss_skip
    logits = out['logits'].reshape(B, S, ids.shape[-1], -1)
    ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), ids.reshape(-1), reduction='none',
                         label_smoothing=float(cfg.label_smoothing)).reshape(B, S, -1)      # :420-423 (smoothed form :99-104)
    ce = ce[:, skip:].mean((1, 2))                                                      # :424-426
    loss = ce * cfg.image_generation_weight                                             # :428
    metrics = dict(ce_loss=ce.mean())
    if cfg.use_localization:                                                            # :430-448
        hidden = out['hidden_states'][-1]                                               # the LOC stream
        raw = mg.mlp(sd, 'pose_criterion.pose_classifier', hidden, dtype)               # :157
        xyz = raw[..., :3]
        if pose_factors is not None:
            xyz = xyz / pose_factors.to(dtype).view(B, 1, 1, 1)                          # :160-161
        y = poses.to(dtype).unsqueeze(-2) * torch.tensor([cfg.pose_multiplier] * 3 + [1.0] * 4, dtype=dtype)   # :167
        pos = ((y[..., :3] - xyz) ** 2).mean(-1)[:, skip:].mean((1, 2))                 # :170-175
        ori = ((y[..., 3:] - raw[..., 3:]) ** 2).mean(-1)[:, skip:].mean((1, 2))
        w = schedule_value(cfg.localization_weight, step)                               # :446
        if cfg.use_dynamic_pose_loss:                                                   # DynamicLossWeightingCriterion :107-120
            sw = sd['pose_loss_weighting_criterion.pos_ori_weights'].to(dtype)
            pose_loss = (sw + torch.exp(-sw) * torch.stack([pos, ori], -1)).sum()       # reduce_SUM over batch and both terms
            metrics['pose_loss'] = pose_loss
        else:
            pose_loss = pos + ori                                                       # :284
        loss = loss + pose_loss * w                                                     # :440,447
        metrics.update(pose_pos_loss=pos.mean(), pose_ori_loss=ori.mean(), localization_weight=w)
    total = loss.mean()                                                                 # :476 reduce_mean
    metrics['loss'] = total
    return total, metrics


def gradients(sd_np, cfg, poses, tokens, step=0, pose_factors=None):
    """fp64 autograd gradients of the restated loss w.r.t. every variable"""
    sd = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in sd_np.items()}
    total, metrics = losses(sd, cfg, torch.as_tensor(poses), torch.as_tensor(tokens), step, pose_factors=pose_factors)
    total.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}
    return grads, {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in metrics.items()}


def learning_rate(step, init_lr, total_steps, warmup_steps, offset=0):
    """WarmUp over CosineDecay (models/utils.py:346-361,403-412); offset: WarmUp.offset (:337,341, set by finetune_transformer.py:86)"""
    step = max(step - offset, 0)
    if warmup_steps and step < warmup_steps:
        return init_lr * (step / warmup_steps)
    decay_steps = max(total_steps - warmup_steps, 1)
    t = min(step - warmup_steps, decay_steps)
    return init_lr * 0.5 * (1.0 + math.cos(math.pi * t / decay_steps))


def process_batch_np(cameras, augment, split, draws=None):
    """fp64 numpy restatement of process_batch (viewformer/train/train_transformer.py:28-61) for ONE sequence ``cameras`` [S,7]; ``draws``: the
    random numbers of the call — shift [3], y0, x, y1 scalars (the reference draws them from tf.random: shapes (1,3) / (1,))."""
    def qmul(a, b):                                   # geometry_tf.py:6-13
        w1, x1, y1, z1 = np.moveaxis(a, -1, 0)
        w2, x2, y2, z2 = np.moveaxis(b, -1, 0)
        return np.stack((-x1 * x2 - y1 * y2 - z1 * z2 + w1 * w2, x1 * w2 + y1 * z2 - z1 * y2 + w1 * x2,
                         -x1 * z2 + y1 * w2 + z1 * x2 + w1 * y2, x1 * y2 - y1 * x2 + z1 * w2 + w1 * z2), -1)

    def qconj(q):                                     # geometry_tf.py:53-56
        return np.concatenate([q[..., :1], -q[..., 1:]], -1)

    def qrot(p, q):                                   # geometry_tf.py:59-68
        p4 = np.concatenate([np.zeros_like(p[..., :1]), p], -1)
        return qmul(qmul(q, p4), qconj(q))[..., 1:]

    def mkq(axis, angle):                             # geometry_tf.py:16-33
        return np.concatenate([[np.cos(angle / 2)], np.sin(angle / 2) * np.asarray(axis, np.float64)])[None]

    cam = np.asarray(cameras, np.float64)
    xyz, q = cam[:, :3], cam[:, 3:]
    if augment == 'relative':                         # :31-36
        rinv = qconj(q[:1])
        xyz = qrot(xyz - xyz[:1], rinv)
        q = qmul(rinv, q)
    elif augment == 'no' or split != 'train':         # :37-38
        pass
    elif augment == 'simple':                         # :39-50
        xyz = xyz + np.asarray(draws['shift'], np.float64)[None]
        rot = qmul(mkq([0, 1, 0], draws['y0']), qmul(mkq([1, 0, 0], draws['x']), mkq([0, 1, 0], draws['y1'])))
        xyz = qrot(xyz, rot)
        q = qmul(q, rot)
    elif augment == 'advanced':                       # :51-55
        xyz = xyz + np.asarray(draws['shift'], np.float64)[None]
        rot = mkq([0, 1, 0], draws['y0'])
        xyz = qrot(xyz, rot)
        q = qmul(q, rot)
    else:
        raise ValueError(f'Augment {augment} is not supported')
    q = q / np.sqrt(np.maximum((q * q).sum(-1, keepdims=True), 1e-12))       # quaternion_normalize, geometry_tf.py:44-45
    q = q * (2.0 * (q[:, :1] >= 0) - 1.0)                                     # quaternion_remove_sign, :48-50
    return np.concatenate([xyz, q], -1)


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


# ---------------------------------------------------------------------------------------------------------------- dropout
def dropout_hash(seed, site, idx):
    """numpy restatement of vf_dropout_hash (viewformer_amd/csrc/vf_common.h): uint32 arithmetic with wrap-around"""
    idx = np.asarray(idx, dtype=np.uint64)
    M = np.uint64(0xFFFFFFFF)
    mul = lambda a, c: (a * np.uint64(c)) & M
    h = np.uint64((int(seed) ^ ((int(site) * 0x9E3779B9) & 0xFFFFFFFF)) & 0xFFFFFFFF)
    h = h ^ (idx & M)
    h = mul(h, 0x85EBCA6B)
    h = h ^ (h >> np.uint64(13))
    h = (h + mul(idx >> np.uint64(32), 0xC2B2AE35) + np.uint64(0x27D4EB2F)) & M
    h = h ^ (h >> np.uint64(16))
    h = mul(h, 0x165667B1)
    h = h ^ (h >> np.uint64(15))
    h = mul(h, 0xD3A2646C)
    h = h ^ (h >> np.uint64(16))
    return h.astype(np.uint32)


def lowbias32(x):
    """numpy restatement of vf_lowbias32 (csrc/vf_common.h): uint32 arithmetic with wrap-around"""
    M = np.uint64(0xFFFFFFFF)
    x = np.asarray(x, dtype=np.uint64) & M
    x = x ^ (x >> np.uint64(16))
    x = (x * np.uint64(0x7FEB352D)) & M
    x = x ^ (x >> np.uint64(15))
    x = (x * np.uint64(0x846CA68B)) & M
    x = x ^ (x >> np.uint64(16))
    return x


def dropout_keep(seed, site, group, sub, thresh):
    """numpy restatement of vf_dropout_keep (csrc/vf_common.h): the four elements of 64-bit group index ``group`` share
    word = lowbias32(lo32(group) ^ vf_dropout_hash(seed, site, hi32(group))); element ``sub`` (0..3) is kept iff rotl32(word, 8 sub) >= thresh"""
    M = np.uint64(0xFFFFFFFF)
    g = np.asarray(group, dtype=np.uint64)
    hi = g >> np.uint64(32)
    key = np.zeros(g.shape, dtype=np.uint64)
    for h in np.unique(hi).tolist():                          # (one key per plane)
        key[hi == np.uint64(h)] = np.uint64(int(dropout_hash(seed, site, np.asarray([h], dtype=np.uint64))[0]))
    w = lowbias32((g & M) ^ key)
    sh = (np.asarray(sub, dtype=np.uint64) & np.uint64(3)) * np.uint64(8)
    rot = np.where(sh == 0, w, ((w << sh) | (w >> (np.uint64(32) - sh))) & M)
    return rot >= np.uint64(thresh)


class DropoutMasks:
    """the masks MIGTTrainer applies (train.py: SITE_EMBED, site_attn/resid/mlp; group / position conventions of csrc/vf_common.h) in the
    shapes the restated forward of migt_oracle needs.  Hidden states are [M = B*V*L][d] row-major in the build (V = NS*S), a list of
    NS tensors [B, S, L, d] here: element (row m, column c) sits in group (m >> 2) * d + c at position m & 3.  Attention weights are indexed
    by absolute (query, key) token positions in the T = V*L sequence: plane b*H + h, group q * ceil(T/4) + (k >> 2), position k & 3."""

    def __init__(self, rate, seed, B, NS, S, L, d, H, dtype=torch.float64, b0=0, pruned_layer=None):
        self.rate, self.seed, self.B, self.NS, self.S, self.L, self.d, self.H, self.dtype = rate, seed, B, NS, S, L, d, H, dtype
        # MIGTTrainer.prune_last_block (round 6): from the last block's projection on, the build runs on the NS - 1 branch streams' rows only
        # (gathered: [B][NS - 1][S][L]) and indexes that block's 'resid' and 'mlp' masks by the gathered rows; the main stream's rows of that
        # block reach no loss, so whatever mask they get here is without effect
        self.pruned_layer = pruned_layer
        self.b0 = b0                                               # index of the first scene in the global batch (MIGTTrainer.scene_offset)
        self.thresh = int(rate * 4294967296.0)
        self.scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(rate)))      # fp32 like the kernels
        self._attn_cache = {}

    def _mask(self, site, group, sub):
        keep = dropout_keep(self.seed, site, group, sub, self.thresh)
        return torch.from_numpy(keep.astype(np.float64) * self.scale).to(self.dtype)

    def elem(self, kind, layer, stream, x):
        site = {'embed': 1, 'resid': 17 + 4 * layer, 'mlp': 18 + 4 * layer}[kind]
        B, S, L, d = self.B, self.S, self.L, self.d
        V = self.NS * S
        if self.pruned_layer is not None and layer == self.pruned_layer and kind in ('resid', 'mlp') and stream >= 1:
            V, stream = (self.NS - 1) * S, stream - 1                  # row index among the gathered branch-stream rows
        b, i, t, c = np.meshgrid(np.arange(B), np.arange(S), np.arange(L), np.arange(d), indexing='ij')
        m = (((b.astype(np.uint64) + np.uint64(self.b0)) * np.uint64(V) + np.uint64(stream * S) + i.astype(np.uint64)) * np.uint64(L)
             + t.astype(np.uint64))
        return self._mask(site, (m >> np.uint64(2)) * np.uint64(d) + c.astype(np.uint64), m & np.uint64(3)).reshape(x.shape)

    def _full(self, layer):
        if layer not in self._attn_cache:
            B, H = self.B, self.H
            T = self.NS * self.S * self.L
            b, h, q, k = np.meshgrid(np.arange(B), np.arange(H), np.arange(T), np.arange(T), indexing='ij')
            k = k.astype(np.uint64)
            group = (((b.astype(np.uint64) + np.uint64(self.b0)) * np.uint64(H) + h.astype(np.uint64)) << np.uint64(32)) | (q.astype(np.uint64) * np.uint64((T + 3) // 4) + (k >> np.uint64(2)))
            self._attn_cache = {layer: self._mask(16 + 4 * layer, group, k & np.uint64(3))}
        return self._attn_cache[layer]

    def attn(self, layer, stream, kind):
        full = self._full(layer)                                   # [B, H, T, T] over absolute token positions
        S, L = self.S, self.L
        r0 = stream * S * L
        rows = full[:, :, r0:r0 + S * L]
        if kind == 'main':
            return rows[:, :, :, :S * L]
        # branch stream: keys = main views 0..S-2, then the query's own view tokens (branching_attention.py:96-124)
        old = rows[:, :, :, :(S - 1) * L]
        q = torch.arange(S * L)
        own_cols = (r0 + (q // L) * L)[:, None] + torch.arange(L)[None, :]            # [S*L, L] absolute key index
        own = torch.gather(rows, 3, own_cols[None, None].expand(rows.shape[0], rows.shape[1], -1, -1))
        return torch.cat([old, own], -1)


def losses_with_dropout(sd, cfg, poses, tokens, step, rate, seed, dtype=torch.float64, pruned_last_block=True):
    B, S = tokens.shape[:2]
    L = int(np.prod(tokens.shape[2:]))
    NS = 3 if cfg.use_localization else 2
    pruned = cfg.n_layer - 1 if (pruned_last_block and cfg.n_layer > 0) else None
    with mg.dropout_masks(DropoutMasks(rate, seed, B, NS, S, L, cfg.d_model, cfg.n_head, dtype, pruned_layer=pruned)):
        return losses(sd, cfg, poses, tokens, step, dtype)


def gradients_with_dropout(sd_np, cfg, poses, tokens, step, rate, seed, pruned_last_block=True):
    sd = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in sd_np.items()}
    total, metrics = losses_with_dropout(sd, cfg, torch.as_tensor(poses), torch.as_tensor(tokens), step, rate, seed, pruned_last_block=pruned_last_block)
    total.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in sd.items()}
    return grads, {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in metrics.items()}
