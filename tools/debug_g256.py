"""where does the 256-tile GEMM differ from the 128-tile one? (debug helper)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewformer_amd import ops
dev = torch.device('cuda:0')
for (M, K, N) in ((512, 128, 256), (512, 256, 256), (1024, 768, 768)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
    w = (torch.randn(K, N, generator=g) * 0.05).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    wp = ops.pack_dense_kn_bf16(w)
    outs = []
    for flag in ('1', '0', '1'):
        os.environ['VF_GEMM_G256'] = flag
        out = torch.full((M, N), float('nan'), device=dev)
        ops.igemm(x, wp, M, K, N, out, bias=b, bf16=True, a16=True)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    ref = (x.double().cpu() @ w.to(torch.bfloat16).double().cpu() + b.double().cpu())
    for name, o in (('g256', outs[0]), ('t128', outs[1]), ('g256 again', outs[2])):
        print(M, K, N, name, 'max err vs fp64', (o.double() - ref).abs().max().item())
    d = (outs[0] != outs[1])
    print('  mismatches', int(d.sum()), 'of', d.numel(), ' rerun-equal', torch.equal(outs[0], outs[2]))
    if d.any():
        rows = d.any(1).nonzero().flatten().numpy(); cols = d.any(0).nonzero().flatten().numpy()
        print('  rows', rows[:40], '... n', len(rows)); print('  cols', cols[:40], '... n', len(cols))
        i, j = d.nonzero()[0].tolist()
        print('  first', i, j, outs[0][i, j].item(), outs[1][i, j].item(), ref[i, j].item())
