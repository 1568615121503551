"""time the 256-tile bf16 GEMM's epilogue forms at the training shape (19 200 x 768 -> 3072): plain, GELU, dual, GELU-backward"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewformer_amd import ops

dev = torch.device('cuda:0')
M, K, N = 19200, 768, 3072
g = torch.Generator(device='cpu').manual_seed(0)
x16 = torch.randn((M, K), generator=g).to(dev).to(torch.bfloat16)
w = (torch.randn((K, N), generator=g) * 0.05).to(dev)
b = torch.randn(N, generator=g).to(dev)
wp = ops.pack_dense_kn_bf16(w)
u32 = torch.randn((M, N), generator=g).to(dev)
u16 = u32.to(torch.bfloat16)
o32 = torch.empty((M, N), device=dev)
o16 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
f16 = torch.empty_like(o16)
cases = {
    'fp32 out': lambda: ops.igemm(x16, wp, M, K, N, o32, bias=b, bf16=True, a16=True),
    'bf16 out': lambda: ops.igemm(x16, wp, M, K, N, o16, bias=b, bf16=True, a16=True, o16=True),
    'bf16 out + GELU': lambda: ops.igemm(x16, wp, M, K, N, o16, bias=b, epilogue=ops.EPI_GELU, bf16=True, a16=True, o16=True),
    'dual (fp32 u + bf16 f)': lambda: ops.igemm(x16, wp, M, K, N, o32, bias=b, epilogue=ops.EPI_GELU_DUAL, bf16=True, a16=True, out_aux=f16),
    'dual (bf16 u + bf16 f)': lambda: ops.igemm(x16, wp, M, K, N, o16, bias=b, epilogue=ops.EPI_GELU_DUAL, bf16=True, a16=True, o16=True, out_aux=f16),
    'GELU-backward, fp32 u': lambda: ops.igemm(x16, wp, M, K, N, o16, res=u32, epilogue=ops.EPI_GELU_BWD, bf16=True, a16=True, o16=True),
    'GELU-backward, bf16 u': lambda: ops.igemm(x16, wp, M, K, N, o16, res=u16, epilogue=ops.EPI_GELU_BWD, bf16=True, a16=True, o16=True, res16=True),
    "dual, saved derivative (bf16 gelu'(u) + bf16 f)": lambda: ops.igemm(x16, wp, M, K, N, o16, bias=b, epilogue=ops.EPI_GELU_DUAL, bf16=True, a16=True, o16=True, out_aux=f16, gelu_grad=True),
    "GELU-backward on the saved derivative": lambda: ops.igemm(x16, wp, M, K, N, o16, res=u16, epilogue=ops.EPI_GELU_BWD, bf16=True, a16=True, o16=True, res16=True, gelu_grad=True),
}
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print(f'{name:28s} {us:7.1f} us   {2.0 * M * K * N / us / 1e6:6.0f} TF')
