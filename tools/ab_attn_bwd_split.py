#!/usr/bin/env python
"""In-process A/B of the bf16 training attention backward, the dQ launch and the dK / dV launch timed SEPARATELY, across builds of libvf_hip.so at the
training step's shape (10 scenes x 12 heads x 3 streams x 10 views), dropout 0 and 0.1, alternated.  Timing only (ablation builds give wrong results);
with AB_CHECK=1 every build's dq / dk / dv is compared with the first build's.  usage: python tools/ab_attn_bwd_split.py lib1.so lib2.so ..."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from viewformer_amd import _lib  # noqa: E402
from viewformer_amd import train_ops as T  # noqa: E402
from viewformer_amd.train_ops import _p, _stream, _drop4, check  # noqa: E402

dev = torch.device('cuda:0')
libs = [(os.path.basename(p), _lib.load_variant(p)) for p in sys.argv[1:]]
B, H, S, L = int(os.environ.get('AB_B', 10)), 12, 30, 64
d, Tn = H * 64, S * L
g = torch.Generator().manual_seed(7)
qkv = (torch.randn(B * Tn, 3 * d, generator=g) * 0.3).to(dev).to(torch.bfloat16)
dout = (torch.randn(B * Tn, d, generator=g) * 0.1).to(dev).to(torch.bfloat16)
q, k, v = qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d]
res = {'shape_B_H_S_L': [B, H, S, L]}
for drop in ((0.0, 0, 0), (0.1, 17, 5)):
    times = {n: {'dq': [], 'dkv': []} for n, _ in libs}
    outs = {}
    for r_ in range(8):
        for n, h in libs:
            with _lib.use(h):
                lib = _lib.load()
                o = torch.empty(B * Tn, d, device=dev, dtype=torch.bfloat16)
                dqkv = torch.zeros(B * Tn, 3 * d, device=dev, dtype=torch.bfloat16)
                lse = T.attn_fwd_lse_bf16(q, k, v, o, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, -10, drop)
                D = torch.empty((B, H, Tn), dtype=torch.float32, device=dev)
                check(lib.vf_attn_bwd_prep_bf16(_p(dout), _p(o), _p(D), B, H, Tn, d, d, _stream()), 'prep')
                dq_, dk_, dv_ = dqkv[:, d:2 * d], dqkv[:, 2 * d:], dqkv[:, :d]

                def launch(a, b_, c):
                    check(lib.vf_attn_bwd_bf16(_p(q), _p(k), _p(v), _p(dout), _p(lse), _p(D), _p(a) if a is not None else None,
                                               _p(b_) if b_ is not None else None, _p(c) if c is not None else None, 1, B, H, Tn, L, 3 * d, 3 * d,
                                               3 * d, d, 3 * d, 3 * d, 3 * d, 1.0, -10, *_drop4(drop), _stream()), 'bwd')
                for which, fn in (('dq', lambda: launch(dq_, None, None)), ('dkv', lambda: launch(None, dk_, dv_))):
                    fn()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    times[n][which].append(e0.elapsed_time(e1) / 10 * 1e3)
                outs[n] = dqkv
    entry = {'us_median': {n: {w: round(statistics.median(t[w]), 1) for w in t} for n, t in times.items()}}
    if os.environ.get('AB_CHECK'):
        first = libs[0][0]
        entry['equal_to_first'] = {n: bool(torch.equal(outs[n], outs[first])) for n in outs}
        entry['max_rel_diff_vs_first'] = {n: float((outs[n].float() - outs[first].float()).abs().max() / outs[first].float().abs().max()) for n in outs}
    res[f'dropout_{drop[0]}'] = entry
print(json.dumps(res), flush=True)
