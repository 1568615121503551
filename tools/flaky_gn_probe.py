"""GroupNorm backward alone under GPU sharing (see profiles/r5_gpu_sharing_transient.txt): N processes loop vf_groupnorm_bwd_f32 (+ the stats of
the forward) on fixed inputs and compare dx / dgamma / dbeta bit for bit with the first call's; on a mismatch: which outputs, which channels / images.
  python tools/flaky_gn_probe.py [calls] [processes] [C] [HW]"""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, n_iter, C, HW):
    from viewformer_amd import ops
    from viewformer_amd import train_ops as T
    dev = torch.device('cuda:0')
    g = np.random.Generator(np.random.PCG64(5 + rank))
    n = 3
    x = torch.from_numpy(g.standard_normal((n * HW, C)).astype(np.float32)).to(dev)
    da = torch.from_numpy(g.standard_normal((n * HW, C)).astype(np.float32)).to(dev)
    gamma = torch.from_numpy((1 + 0.1 * g.standard_normal(C)).astype(np.float32)).to(dev)
    beta = torch.from_numpy((0.1 * g.standard_normal(C)).astype(np.float32)).to(dev)
    filler = torch.randn(2048, 2048, device=dev)
    ref = None
    bad = 0
    for it in range(n_iter):
        mean_c, scale_c = ops.groupnorm_stats(x, gamma, n, HW, C, 32, 1e-6)
        dx, dg, db = T.groupnorm_bwd(x, da, mean_c, scale_c, gamma, beta, n, HW, C, True)
        if it % 7 == 0:
            filler = filler @ filler * 1e-3                       # other kernels of this process in between
        cur = (mean_c.clone(), scale_c.clone(), dx.clone(), dg.clone(), db.clone())
        if ref is None:
            torch.cuda.synchronize()
            ref = cur
            continue
        if it % 50 == 0 or True:
            eq = [torch.equal(a, b) for a, b in zip(cur, ref)]
            if not all(eq):
                bad += 1
                names = ['mean_c', 'scale_c', 'dx', 'dgamma', 'dbeta']
                msg = {nm: 'equal' if e else f'max |diff| {float((a - b).abs().max()):.3e} at {int((a - b).abs().argmax())}' for nm, e, a, b in zip(names, eq, cur, ref)}
                dch = (cur[3] - ref[3]).abs().nonzero().flatten().tolist()
                dxd = (cur[2] - ref[2]).abs().view(n, HW, C).amax(1)
                print(f'rank {rank} call {it}: {msg}; dgamma channels that differ: {dch[:8]}; dx differs in (image, channel): {dxd.nonzero().tolist()[:12]}', flush=True)
    print(f'rank {rank}: {bad} of {n_iter - 1} calls differ', flush=True)


if __name__ == '__main__':
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    nproc = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    C = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    HW = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    mp.set_start_method('spawn')
    ps = [mp.Process(target=worker, args=(r, n_iter, C, HW)) for r in range(nproc)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
