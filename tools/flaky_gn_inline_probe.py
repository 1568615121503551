"""locate the GPU-sharing transient AT ITS SOURCE (profiles/r5_gpu_sharing_transient.txt): N processes run the codebook training step; every
GroupNorm-backward call inside it is issued THREE times on the same inputs and the three results are compared bit for bit on the spot (the
inputs are still alive), so an outlier launch is caught with its inputs, outputs and position.
  python tools/flaky_gn_inline_probe.py [steps] [processes]"""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, n_iter):
    from viewformer_amd import train_ops as T
    from viewformer_amd import vqgan_train as VT
    from viewformer_amd.config import VQGANConfig
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.weights import make_vqgan_weights
    dev = torch.device('cuda:0')
    orig = T.groupnorm_bwd
    stats = dict(calls=0, events=0)

    def checked(x, da, mean_c, scale_c, gamma, beta, n_img, HW, C, swish, groups=32):
        outs = [tuple(t.clone() for t in orig(x, da, mean_c, scale_c, gamma, beta, n_img, HW, C, swish, groups)) for _ in range(3)]
        stats['calls'] += 1
        same01 = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
        same02 = all(torch.equal(a, b) for a, b in zip(outs[0], outs[2]))
        same12 = all(torch.equal(a, b) for a, b in zip(outs[1], outs[2]))
        if not (same01 and same02):
            stats['events'] += 1
            odd = 0 if same12 else 1 if same02 else 2 if same01 else -1
            good = outs[(odd + 1) % 3] if odd >= 0 else outs[0]
            bad = outs[odd] if odd >= 0 else outs[1]
            names = ['dx', 'dgamma', 'dbeta']
            det = {}
            for nm, a, b in zip(names, bad, good):
                if not torch.equal(a, b):
                    d = (a - b).abs()
                    if nm == 'dx':
                        dd = d.view(n_img, HW, C)
                        det[nm] = dict(images=dd.amax((1, 2)).nonzero().flatten().tolist(), channels=dd.amax((0, 1)).nonzero().flatten().tolist()[:16],
                                       pixels=int((dd.amax(2) > 0).sum()), max=float(d.max()))
                    else:
                        idx = d.nonzero().flatten().tolist()
                        det[nm] = dict(channels=idx[:16], bad=[float(a[i]) for i in idx[:4]], good=[float(b[i]) for i in idx[:4]])
            print(f'rank {rank} call {stats["calls"]}: outlier launch #{odd} of 3; C={C} HW={HW} n={n_img} swish={swish}; {det}', flush=True)
        return outs[0] if same01 or same02 else outs[1]
    T.groupnorm_bwd = checked
    VT.T.groupnorm_bwd = checked
    cfg = VQGANConfig(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], image_size=32, z_channels=32, embed_dim=32, n_embed=64,
                      perceptual_weight=0.0, codebook_weight=1.0, learning_rate=1e-3)
    sd = make_vqgan_weights(cfg, seed=3, codebook_scale=0.05)
    g = np.random.Generator(np.random.PCG64(50 + rank))
    img = torch.from_numpy((g.random((3, 3, 32, 32)) * 2 - 1).astype(np.float32))
    for it in range(n_iter):
        model = VQGAN(cfg, device=dev)
        model.load_state_dict(sd)
        tr = VT.VQGANTrainer(model)
        tr.train_step(img, reduce_gradients=False, apply_update=False)
        if it % 200 == 199:
            print(f'rank {rank}: {it + 1} steps, {stats["calls"]} GroupNorm-backward calls (x3 launches), {stats["events"]} events', flush=True)


if __name__ == '__main__':
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    nproc = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    mp.set_start_method('spawn')
    ps = [mp.Process(target=worker, args=(r, n_iter)) for r in range(nproc)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
