#!/usr/bin/env python
"""Does the MLP's hidden activation have to go through HBM?  (measured instead of argued.)

c_fc writes the 65 536 x 3072 bf16 hidden (403 MB) and mlp.c_proj reads it back.  A kernel-level fusion cannot hold it: a 256-row panel's
c_proj output (256 x 768 fp32 = 768 KB) is more than a CU's whole register file (512 KB), and with a 128-row panel the fused tile moves as
many operand bytes per flop through the CU as the two 256-tile GEMMs do.  What CAN keep the hidden on chip without a new kernel is
launch order: run c_fc and c_proj back to back on row CHUNKS whose hidden slice fits the 256 MB Infinity Cache.  This script times the
pair at the bench's M for chunk sizes 65 536 (two launches, as the product does), 32 768, 16 384 and 8 192 rows, with the residual add and
the GELU epilogue as in the product path."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewformer_amd import ops

dev = torch.device('cuda:0')
M, d = 65536, 768
x = (torch.randn(M, d, device=dev) * 0.5).to(torch.bfloat16)
wfc = ops.pack_dense_kn_bf16(torch.randn(d, 4 * d, device=dev) * 0.02)
wpr = ops.pack_dense_kn_bf16(torch.randn(4 * d, d, device=dev) * 0.02)
bfc, bpr = torch.randn(4 * d, device=dev), torch.randn(d, device=dev)
res = torch.randn(M, d, device=dev)
hid = torch.empty(M, 4 * d, device=dev, dtype=torch.bfloat16)
out = torch.empty(M, d, device=dev)


def pair(chunk):
    for r0 in range(0, M, chunk):
        n = min(chunk, M - r0)
        ops.igemm(x[r0:r0 + n], wfc, n, d, 4 * d, hid[r0:r0 + n], bias=bfc, epilogue=ops.EPI_GELU, bf16=True, a16=True, o16=True)
        ops.igemm(hid[r0:r0 + n], wpr, n, 4 * d, d, out[r0:r0 + n], bias=bpr, res=res[r0:r0 + n], bf16=True, a16=True)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


fl = 2.0 * M * d * 4 * d * 2
ref = None
for rep in range(2):
    for chunk in (65536, 32768, 16384, 8192):
        us = timeit(lambda: pair(chunk))
        if ref is None:
            pair(65536)
            ref = out.clone()
        pair(chunk)
        same = torch.equal(out, ref)
        print(f'c_fc + mlp.c_proj, M = {M}, chunks of {chunk:6d} rows (hidden slice {chunk * 4 * d * 2 / 2 ** 20:6.0f} MiB): {us:8.1f} us  '
              f'{fl / us / 1e6:7.1f} TF   same bits: {same}')
