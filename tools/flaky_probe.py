"""is the bf16 training step bit-deterministic while ANOTHER process shares the GPU?  (the 2-rank gloo test on one GPU found rare mismatches)
Two processes run the same full-size step over and over; each compares every repeat with its first result and lists the tensors that differ."""
import os
import sys
import torch
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))


def worker(rank, n_iter, flags):
    import test_hip_train_full as F
    dev = torch.device('cuda:0')
    cfg = F._cfg()
    tr = F._trainer(cfg, dev, 'bf16')
    for k, v in flags.items():
        setattr(tr, k, v)
    poses, tok = F._batch(2, 10, 100 + rank)
    tr.train_step(poses, tok, reduce_gradients=False, apply_update=False)
    g0 = tr.flat_g.clone()
    bad = 0
    for it in range(n_iter):
        tr.train_step(poses, tok, reduce_gradients=False, apply_update=False)
        if not torch.equal(tr.flat_g, g0):
            bad += 1
            names = [n for n in tr.names if not torch.equal(tr.flat_g[tr.slices[n][0]:tr.slices[n][1]], g0[tr.slices[n][0]:tr.slices[n][1]])]
            layers = sorted({int(n.split('.')[1]) for n in names if n.startswith('h.')})
            top = max(layers) if layers else -1
            print(f'rank {rank} iter {it}: {len(names)} tensors differ; layers {layers}; in top layer {top}: {[n for n in names if n.startswith("h.%d." % top)]}; '
                  f'head: {[n for n in names if not n.startswith("h.")]}', flush=True)
    print(f'rank {rank}: {bad} of {n_iter} repeats differ  flags={flags}', flush=True)


if __name__ == '__main__':
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    flags = {}
    for a in sys.argv[2:]:
        k, v = a.split('=')
        flags[k] = {'True': True, 'False': False}.get(v, v)
    mp.set_start_method('spawn')
    ps = [mp.Process(target=worker, args=(r, n_iter, flags)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
