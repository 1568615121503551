"""vf_layernorm_bwd_f32 under GPU sharing: two processes repeat the same call and compare every output with their first result"""
import os
import sys
import torch
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, n_iter):
    from viewformer_amd import train_ops as T
    dev = torch.device('cuda:0')
    M, d = 3840, 768
    g = torch.Generator(device='cpu').manual_seed(rank)
    dy, x, res = (torch.randn((M, d), generator=g).to(dev) for _ in range(3))
    gamma = torch.randn(d, generator=g).to(dev)
    filler = torch.randn((4096, 4096), device=dev)

    def call():
        dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        dx, dx16 = T.layernorm_bwd(dy, x, gamma, dg, db, M, d, res=res, also_bf16=True)
        return dx, dx16, dg, db
    ref = [t.clone() for t in call()]
    bad = 0
    for it in range(n_iter):
        _ = filler @ filler                                    # some other work between the calls, as in the step
        out = call()
        torch.cuda.synchronize()
        diff = [not torch.equal(a, b) for a, b in zip(out, ref)]
        if any(diff):
            bad += 1
            rows = (out[0] != ref[0]).any(1).nonzero().flatten().tolist()
            rows16 = (out[1] != ref[1]).any(1).nonzero().flatten().tolist()
            if bad <= 6:
                print(f'rank {rank} iter {it}: differs dx={diff[0]} dx16={diff[1]} dgamma={diff[2]} dbeta={diff[3]}; dx rows {rows[:12]} ({len(rows)}), dx16 rows {rows16[:12]} ({len(rows16)})',
                      flush=True)
    print(f'rank {rank}: {bad} of {n_iter} calls differ (two rows in flight: {os.environ.get("VF_LN_BWD_TWO_ROWS", "1")})', flush=True)


if __name__ == '__main__':
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    mp.set_start_method('spawn')
    ps = [mp.Process(target=worker, args=(r, n_iter)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
