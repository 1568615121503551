#!/usr/bin/env python
"""Isolated timing of the optimizer step's two forms at the training model's size (12 layers of 768 x {2304, 768, 3072} + 3072 x 768, the tied
head 1024 x 768, 3.4 M other parameters): vf_adamw_flat_f32 + vf_gemm_bf16_pack_multi against vf_adamw_flat_pack_f32.
usage (GPU box): python tools/bench_adamw_pack.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from viewformer_amd import ops, train_ops as T  # noqa: E402

dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = []
for _ in range(12):
    shapes += [(768,), (768,), (768, 2304), (2304,), (768, 768), (768,), (768,), (768,), (768, 3072), (3072,), (3072, 768), (768,)]
shapes = [(1026, 768), (64, 768)] + shapes + [(1536, 7), (768,), (768,)]
offs, n = [], 0
for s in shapes:
    offs.append(n)
    numel = 1
    for d in s:
        numel *= d
    n += (numel + 3) // 4 * 4
p = torch.randn(n, device=dev) * 0.02
g = torch.randn(n, device=dev) * 0.01
m = torch.zeros(n, device=dev)
v = torch.zeros(n, device=dev)
L = ops._lib.load()
items, multi = [], []
for s, o in zip(shapes, offs):
    if len(s) == 2 and s[1] % 128 == 0 and s[0] >= 128:
        r = s[0] // 128 * 128
        w = p[o:o + r * s[1]].view(r, s[1])
        kn = torch.empty(int(L.vf_gemm_bf16_packed_elems(r, s[1])), dtype=torch.bfloat16, device=dev)
        nk = torch.empty(int(L.vf_gemm_bf16_packed_elems(s[1], r)), dtype=torch.bfloat16, device=dev)
        items.append((w, kn, nk))
        multi += [(w, False, kn), (w, True, nk)]
table = T.adamw_pack_table(p, items)
repack = ops.pack_bf16_multi(multi)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


args = (None, 0.0, 1e-4, 0.9, 0.999, 1e-7)
print(f'parameters {n / 1e6:.1f} M, packed matrices {len(items)}')
print(f'adamw_flat            {timed(lambda: T.adamw_flat_(p, g, m, v, *args)):8.1f} us')
print(f'pack_bf16_multi       {timed(repack):8.1f} us')
print(f'adamw_flat_pack       {timed(lambda: T.adamw_flat_pack_(p, g, m, v, *args, table)):8.1f} us')
