"""locate the nondeterministic op of the bf16 training step under GPU sharing: every layernorm_bwd call's inputs / outputs are compared with
the first step's (two processes on one GPU)"""
import os
import sys
import torch
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))


def worker(rank, n_iter):
    import test_hip_train_full as F
    from viewformer_amd import train_ops as T
    dev = torch.device('cuda:0')
    tr = F._trainer(F._cfg(), dev, 'bf16')
    poses, tok = F._batch(2, 10, 100 + rank)
    orig = T.layernorm_bwd
    rec, state = [], dict(step=0, idx=0, reported=False)

    def wrapped(dy, x, gamma, dgamma, dbeta, rows, d, eps=1e-5, accumulate=True, res=None, also_bf16=False):
        out = orig(dy, x, gamma, dgamma, dbeta, rows, d, eps=eps, accumulate=accumulate, res=res, also_bf16=also_bf16)
        dx, dx16 = out if also_bf16 else (out, None)
        cur = dict(dy=dy, x=x, res=res, dx=dx, dx16=dx16)
        if state['step'] == 0:
            rec.append({k: (v.clone() if v is not None else None) for k, v in cur.items()})
        elif not state['reported']:
            r = rec[state['idx']]
            eq = {k: (v is None or torch.equal(v, r[k])) for k, v in cur.items()}
            if not all(eq.values()):
                state['reported'] = True
                bad = [k for k, v in eq.items() if not v]
                msg = f'rank {rank} step {state["step"]} ln_bwd call {state["idx"]}: differs in {bad}'
                for k in bad:
                    a, b = cur[k].float(), r[k].float()
                    rows_bad = (a != b).any(1).nonzero().flatten().tolist()
                    msg += f'; {k}: {len(rows_bad)} rows, first {rows_bad[:8]}, max |diff| {float((a - b).abs().max()):.3e}'
                print(msg, flush=True)
        state['idx'] += 1
        return out
    T.layernorm_bwd = wrapped
    import viewformer_amd.train as TR
    TR.T.layernorm_bwd = wrapped
    for it in range(n_iter + 1):
        state['idx'] = 0
        state['reported'] = False
        tr.train_step(poses, tok, reduce_gradients=False, apply_update=False)
        state['step'] += 1
    print(f'rank {rank} done', flush=True)


if __name__ == '__main__':
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    mp.set_start_method('spawn')
    ps = [mp.Process(target=worker, args=(r, n_iter)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
