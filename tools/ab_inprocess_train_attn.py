#!/usr/bin/env python
"""In-process A/B of the bf16 training attention (forward with log-sum-exp, dQ, dK/dV) across builds of libvf_hip.so at the training step's
shape (10 scenes x 12 heads x 3 streams x 10 views), dropout 0 and 0.1, timed in alternation; accuracy of each build's dQ / dK / dV against fp64
autograd of the reference formula on one scene (dropout 0).  usage: python tools/ab_inprocess_train_attn.py lib1.so lib2.so ..."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from viewformer_amd import _lib  # noqa: E402
from viewformer_amd import train_ops as T  # noqa: E402

dev = torch.device('cuda:0')
libs = [(os.path.basename(p), _lib.load_variant(p)) for p in sys.argv[1:]]
B, H, S, L = 10, 12, 30, 64
d, Tn = H * 64, S * L
g = torch.Generator().manual_seed(7)
SCALE = float(os.environ.get('AB_QSCALE', '0.3'))
qkv = (torch.randn(B * Tn, 3 * d, generator=g) * SCALE).to(dev).to(torch.bfloat16)
dout = (torch.randn(B * Tn, d, generator=g) * 0.1).to(dev).to(torch.bfloat16)
q, k, v = qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d]


def run(drop):
    o = torch.empty(B * Tn, d, device=dev, dtype=torch.bfloat16)
    dqkv = torch.zeros(B * Tn, 3 * d, device=dev, dtype=torch.bfloat16)
    lse = T.attn_fwd_lse_bf16(q, k, v, o, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, -10, drop)
    T.attn_bwd_bf16(q, k, v, o, dout, lse, dqkv[:, d:2 * d], dqkv[:, 2 * d:], dqkv[:, :d], B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d,
                    1.0, -10, drop)
    return o, lse, dqkv


def reference_scene0():
    """fp64 autograd of softmax(q k^T masked) v on scene 0 with the 3-stream mask (branching_attention.py:82-126)"""
    Sv = 10
    x = qkv[:Tn].double().cpu().view(Tn, 3, H, 64)
    vv, qq, kk = (x[:, i].permute(1, 0, 2).clone().requires_grad_(True) for i in range(3))
    view = torch.arange(Tn) // L
    qs, qi = view // Sv, view % Sv
    ks, ki = qs, qi
    vis = torch.where(qs[:, None] == 0, (ks[None, :] == 0) & (ki[None, :] <= qi[:, None]),
                      ((ks[None, :] == 0) & (ki[None, :] < qi[:, None])) | (view[None, :] == view[:, None]))
    s = qq @ kk.transpose(1, 2)
    s = s.masked_fill(~vis[None], float('-inf'))
    o = torch.softmax(s, -1) @ vv
    o.backward(dout[:Tn].double().cpu().view(Tn, H, 64).permute(1, 0, 2))
    return o.detach(), qq.grad, kk.grad, vv.grad


ref_o, ref_dq, ref_dk, ref_dv = reference_scene0()


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


res = {'shape_B_H_S_L': [B, H, S, L], 'q_scale': SCALE, 'max_abs_score': float((ref_o * 0).sum()) if False else None}
for drop in ((0.0, 0, 0), (0.1, 17, 5)):
    times = {n: {'fwd': [], 'bwd': []} for n, _ in libs}
    outs = {}
    for n, h in libs:
        with _lib.use(h):
            outs[n] = run(drop)
    torch.cuda.synchronize()
    for r_ in range(8):
        for n, h in libs:
            with _lib.use(h):
                o = torch.empty(B * Tn, d, device=dev, dtype=torch.bfloat16)
                dqkv = torch.zeros(B * Tn, 3 * d, device=dev, dtype=torch.bfloat16)
                lse = T.attn_fwd_lse_bf16(q, k, v, o, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, -10, drop)
                torch.cuda.synchronize()
                for which, fn in (('fwd', lambda: T.attn_fwd_lse_bf16(q, k, v, o, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, -10, drop)),
                                  ('bwd', lambda: T.attn_bwd_bf16(q, k, v, o, dout, lse, dqkv[:, d:2 * d], dqkv[:, 2 * d:], dqkv[:, :d], B, H, Tn, L,
                                                                  3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, -10, drop))):
                    fn()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    times[n][which].append(e0.elapsed_time(e1) / 10 * 1e3)
    entry = {'us_median': {n: {w: round(statistics.median(t[w]), 1) for w in t} for n, t in times.items()}}
    first = libs[0][0]
    entry['max_rel_diff_vs_first'] = {n: {'out': rel(outs[n][0].float(), outs[first][0].float()), 'dqkv': rel(outs[n][2].float(), outs[first][2].float())}
                                      for n in outs}
    if drop[0] == 0.0:
        acc = {}
        for n in outs:
            o, lse, dqkv = outs[n]
            x = dqkv[:Tn].float().cpu().double().view(Tn, 3, H, 64)
            dv_, dq_, dk_ = (x[:, i].permute(1, 0, 2) for i in range(3))
            acc[n] = {'out': rel(o[:Tn].float().cpu().double().view(Tn, H, 64).permute(1, 0, 2), ref_o), 'dq': rel(dq_, ref_dq), 'dk': rel(dk_, ref_dk),
                      'dv': rel(dv_, ref_dv)}
        entry['max_rel_err_vs_fp64_autograd_scene0'] = acc
    res[f'dropout_{drop[0]}'] = entry
res.pop('max_abs_score')
print(json.dumps(res), flush=True)
