"""the 256-tile LDS-DMA bf16 GEMM across reduction depths: is the distance to the guide's 8-phase figure (1.3 PF at 4096^3) the K-loop
schedule or the per-tile prologue / epilogue at K = 768?  (bf16 in, bf16 out, no bias; random operands)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewformer_amd import ops

dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)


def run(M, K, N, out16=True):
    x16 = torch.randn((M, K), generator=g).to(dev).to(torch.bfloat16)
    wp = ops.pack_dense_kn_bf16((torch.randn((K, N), generator=g) * 0.05).to(dev))
    o = torch.empty((M, N), dtype=torch.bfloat16 if out16 else torch.float32, device=dev)
    fn = lambda: ops.igemm(x16, wp, M, K, N, o, bf16=True, a16=True, o16=out16)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    tiles = ((M + 255) // 256) * (N // 256)
    print(f'M={M:6d} K={K:5d} N={N:5d}  {us:8.1f} us  {2.0 * M * K * N / us / 1e6:6.0f} TF   tiles={tiles} ({tiles / 256:.2f} rounds of 256 CUs)  '
          f'us per tile-round={us / -(-tiles // 256):.1f}')


for shape in ((65536, 128, 3072), (65536, 256, 3072), (65536, 512, 3072), (65536, 768, 3072), (65536, 1536, 3072), (65536, 3072, 3072), (65536, 6144, 3072),
              (4096, 4096, 4096), (8192, 8192, 8192), (16384, 4096, 4096), (65536, 3072, 768), (65536, 768, 768), (65536, 768, 2304),
              (19200, 768, 3072), (19200, 768, 2304), (19200, 3072, 768)):
    run(*shape)
for shape in ((65536, 128, 3072), (65536, 768, 3072), (65536, 3072, 768)):
    run(*shape, out16=False)
