#!/usr/bin/env python
"""In-process A/B of the LayerNorm backward (training shape: 19 200 rows x 768, residual joined, bf16 copy with dropout mask) across builds.
usage: python tools/ab_inprocess_ln_bwd.py lib1.so lib2.so ..."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from viewformer_amd import _lib  # noqa: E402
from viewformer_amd import train_ops as T  # noqa: E402

dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
libs = [(os.path.basename(p), _lib.load_variant(p)) for p in sys.argv[1:]]
rows, d = 19200, 768
g = torch.Generator().manual_seed(3)
dy, x, res = (torch.randn(rows, d, generator=g).to(dev) for _ in range(3))
gamma = torch.randn(d, generator=g).to(dev)


def run():
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    return T.layernorm_bwd(dy, x, gamma, dg, db, rows, d, res=res, also_bf16=True, drop=(0.1, 5, 2)), dg, db


times = {n: [] for n, _ in libs}
outs = {}
for r in range(10):
    for n, h in (libs if r % 2 == 0 else libs[::-1]):
        with _lib.use(h):
            for _ in range(3):
                o = run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                o = run()
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / 20 * 1e3)
            outs[n] = o
first = libs[0][0]
same = {n: bool(torch.equal(outs[n][0][0], outs[first][0][0])) for n in outs}
print(json.dumps({'us_median_incl_two_small_fills_and_final': {n: round(statistics.median(t), 2) for n, t in times.items()}, 'dx_same_bits_as_first': same}))
