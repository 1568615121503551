import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from viewformer_amd import ops
dev = torch.device('cuda:0')
scale, M = 1.0, 200000
g = np.random.Generator(np.random.PCG64(int(scale * 1000) + M))
z = torch.from_numpy((g.standard_normal((M, 256)) * scale).astype(np.float32)).to(dev)
g2 = np.random.Generator(np.random.PCG64(1))
E = torch.from_numpy(((g2.random((256, 1024)) * 2 - 1) * np.sqrt(3.0) * 0.05).astype(np.float32)).to(dev)
Ep, esq = ops.vq_pack_codebook(E)
exact = ops.vq_argmin(z, Ep, esq, 256, 1024)
blob = ops.vq_filter_pack(E)
for it in range(3):
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    filt = ops.vq_argmin_filtered(z, blob, 256, 1024, stats=st)
    bad = (exact != filt).nonzero().reshape(-1).cpu().numpy()
    print('iter', it, 'mismatches', len(bad), st.cpu().tolist())
    for i in bad[:10]:
        zi = z[i].double().cpu().numpy(); En = E.double().cpu().numpy()
        d = (zi * zi).sum() - 2 * zi @ En + (En * En).sum(0)
        o = np.argsort(d)[:4]
        z16 = z[i].half().double().cpu().numpy(); E16 = E.half().double().cpu().numpy()
        s16 = z16 @ E16 - 0.5 * (En * En).sum(0)
        o16 = np.argsort(-s16)[:5]
        print(' row', i, 'wave-row', i % 32, 'blk', i // 128, 'exact', int(exact[i]), 'filt', int(filt[i]), 'fp64 top4', o.tolist(), (d[o] - d[o[0]]).tolist(),
              'filter top5', o16.tolist(), (s16[o16[0]] - s16[o16]).tolist(), 'lanes', (o16 % 32).tolist())
