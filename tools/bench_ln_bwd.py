"""time vf_layernorm_bwd_f32 at the training shape (19 200 x 768, residual + bf16 copy) — A/B of the rows-in-flight forms (vf_select(VF_SEL_LN_BWD_TWO_ROWS, .))"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewformer_amd import train_ops as T
from viewformer_amd import _lib

dev = torch.device('cuda:0')
M, d = 19200, 768
g = torch.Generator(device='cpu').manual_seed(0)
dy, x, res = (torch.randn((M, d), generator=g).to(dev) for _ in range(3))
gamma = torch.randn(d, generator=g).to(dev)
out = {}
for r1 in ('1', '0'):
    _lib.select(_lib.SEL_LN_BWD_TWO_ROWS, r1 == '0')          # r1 = '1': the one-row form
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    for _ in range(5):
        dx, dx16 = T.layernorm_bwd(dy, x, gamma, dg, db, M, d, res=res, also_bf16=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        dx, dx16 = T.layernorm_bwd(dy, x, gamma, dg, db, M, d, res=res, also_bf16=True)
    e1.record()
    torch.cuda.synchronize()
    dg2, db2 = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    dx, dx16 = T.layernorm_bwd(dy, x, gamma, dg2, db2, M, d, res=res, also_bf16=True)
    out[r1] = (dx.clone(),)
    us = e0.elapsed_time(e1) * 1000 / 50
    print(f'one row per wave in flight={r1}: {us:.1f} us per call (kernel + finalize), {265.4e6 / us / 1e6:.2f} TB/s of 265 MB')
print('same bits:', all(torch.equal(a, b) for a, b in zip(out['1'], out['0'])))
