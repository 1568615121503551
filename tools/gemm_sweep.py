#!/usr/bin/env python
"""fixed-vs-per-K cost of the GEMM kernel: time = a + b*K at fixed M x N (GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viewformer_amd import ops
from tools.microbench import timeit
dev = torch.device('cuda:0')
M, N = 8192, 3072
for K in (128, 256, 768, 1536, 3072):
    x = torch.randn(M, K, device=dev)
    wp = ops.pack_dense_kn(torch.randn(K, N, device=dev) * 0.02)
    out = torch.empty(M, N, device=dev)
    ms = timeit(lambda: ops.igemm(x, wp, M, K, N, out), iters=10)
    print(f'K={K:5d}: {ms * 1e3:8.1f} us  {2.0 * M * K * N / ms / 1e9:6.1f} TF', flush=True)
