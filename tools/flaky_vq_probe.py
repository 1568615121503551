"""is the codebook (VQGAN) training step bit-deterministic while ANOTHER process shares the GPU?  tests/test_hip_multirank.py::
test_codebook_trainers_world2_mean_and_ema_allreduce (two ranks on one GPU over gloo) failed once in four runs in round 5 with a 1.8e-3 gradient
mismatch between a trainer's local gradient and a fresh trainer's reduced one.  Two processes build a FRESH trainer per repeat on the same images
and compare the gradient (and the forward's codes / loss) with the first repeat's; differing tensors are listed.
  python tools/flaky_vq_probe.py [repeats] [processes]"""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, n_iter):
    from viewformer_amd.config import VQGANConfig
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.vqgan_train import VQGANTrainer
    from viewformer_amd.weights import make_vqgan_weights
    dev = torch.device('cuda:0')
    cfg = VQGANConfig(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], image_size=32, z_channels=32, embed_dim=32, n_embed=64,
                      perceptual_weight=0.0, codebook_weight=1.0, learning_rate=1e-3)
    sd = make_vqgan_weights(cfg, seed=3, codebook_scale=0.05)
    g = np.random.Generator(np.random.PCG64(50 + rank))
    img = torch.from_numpy((g.random((3, 3, 32, 32)) * 2 - 1).astype(np.float32))
    g0, m0, bad = None, None, 0
    for it in range(n_iter):
        model = VQGAN(cfg, device=dev)
        model.load_state_dict(sd)
        tr = VQGANTrainer(model)
        met = tr.train_step(img, reduce_gradients=False, apply_update=False)
        torch.cuda.synchronize()
        gi = tr.flat_g.clone()
        mi = {k: float(v) for k, v in met.items() if hasattr(v, '__float__')}
        if g0 is None:
            g0, m0 = gi, mi
            continue
        if not torch.equal(gi, g0):
            bad += 1
            names = [(n, float((gi[a:b] - g0[a:b]).abs().max() / g0[a:b].abs().max().clamp_min(1e-30))) for n, (a, b, _) in tr.slices.items()
                     if not torch.equal(gi[a:b], g0[a:b])]
            names.sort(key=lambda t: -t[1])
            dm = {k: (mi[k], m0[k]) for k in mi if mi[k] != m0.get(k)}
            print(f'rank {rank} repeat {it}: {len(names)} of {len(tr.slices)} tensors differ; worst {names[:4]}; metrics that differ: {dm}', flush=True)
            same = [n for n, (a, b, _) in tr.slices.items() if torch.equal(gi[a:b], g0[a:b])]
            dec_diff = [n for n, _ in sorted(names) if n.startswith(('decoder', 'post_quant'))]
            dec_same = [n for n in same if n.startswith(('decoder', 'post_quant'))]
            # backward order = reverse of the forward's parameter order: the LAST name (in state-dict order) that differs is where the error entered
            order = list(tr.slices)
            differing = {n for n, _ in names}
            last = max(i for i, n in enumerate(order) if n in differing)
            print(f'   entered at: {order[last]}  (next in forward order, equal: {order[last + 1:last + 4]})', flush=True)
    print(f'rank {rank}: {bad} of {n_iter - 1} repeats differ from the first', flush=True)


if __name__ == '__main__':
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    nproc = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    mp.set_start_method('spawn')
    ps = [mp.Process(target=worker, args=(r, n_iter)) for r in range(nproc)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
