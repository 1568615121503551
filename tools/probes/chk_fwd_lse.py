"""Forward bf16 attention with log-sum-exp at the training shape through several builds of the library: are out and lse equal bit for bit?
usage: python tools/probes/chk_fwd_lse.py lib1.so lib2.so"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from viewformer_amd import _lib
from viewformer_amd import train_ops as T
dev = torch.device('cuda:0')
libs = [(os.path.basename(p), _lib.load_variant(p)) for p in sys.argv[1:]]
B, H, S, L = 10, 12, 30, 64
d, Tn = H * 64, S * L
g = torch.Generator().manual_seed(7)
qkv = (torch.randn(B * Tn, 3 * d, generator=g) * 0.3).to(dev).to(torch.bfloat16)
q, k, v = qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d]
res = {}
for drop in ((0.0, 0, 0), (0.1, 17, 5)):
    for n, h in libs:
        with _lib.use(h):
            o = torch.empty(B * Tn, d, device=dev, dtype=torch.bfloat16)
            lse = T.attn_fwd_lse_bf16(q, k, v, o, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, -10, drop)
            torch.cuda.synchronize()
            res[(drop[0], n)] = (o.clone(), lse.clone())
    a, b = [res[(drop[0], n)] for n, _ in libs]
    print(drop[0], 'out equal', torch.equal(a[0], b[0]), 'lse equal', torch.equal(a[1], b[1]), 'lse maxdiff', float((a[1] - b[1]).abs().max()),
          'rows differing', int(((a[1] != b[1]).sum())))
    if not torch.equal(a[1], b[1]):
        idx = (a[1] != b[1]).nonzero()
        print(idx[:10].tolist(), a[1].shape)
