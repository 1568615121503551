#!/usr/bin/env python
"""The GPU-sharing transient, one script (replaces round 5's seven flaky_*.py probes; profiles/r5_gpu_sharing_transient.txt has their findings):
N processes time-sliced on ONE GPU run the codebook training step with ``VQGANTrainer.debug_triple_groupnorm_bwd`` on — every GroupNorm backward
is issued three times on the same live inputs and compared bit for bit — for a bounded wall time; every event is printed as one JSON line, then a
summary per process.  The product configuration (one process per GPU) is the N = 1 run.
  python tools/transient_probe.py <seconds> <processes>"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, seconds, q):
    try:
        _worker(rank, seconds, q)
    except Exception as e:      # noqa: BLE001  (the parent must never wait for a dead worker)
        import traceback
        q.put(dict(rank=rank, error=traceback.format_exc() + repr(e)))


def _worker(rank, seconds, q):
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
    t_end = time.time() + seconds
    steps = events = grad_diff = 0
    g0 = None
    while time.time() < t_end:
        model = VQGAN(cfg, device=dev)
        model.load_state_dict(sd)
        tr = VQGANTrainer(model)
        tr.debug_triple_groupnorm_bwd = True
        tr.train_step(img, reduce_gradients=False, apply_update=False)
        torch.cuda.synchronize()
        steps += 1
        for ev in tr.transient_events:
            events += 1
            print(json.dumps(dict(rank=rank, probe_step=steps, t=round(seconds - (t_end - time.time()), 1), **ev)), flush=True)
        # the step's gradient against the first step's (same data, fresh trainer): does anything differ WITHOUT a triple-launch event?
        if g0 is None:
            g0 = tr.flat_g.clone()
        elif not torch.equal(tr.flat_g, g0):
            grad_diff += 1
            if not tr.transient_events:
                print(json.dumps(dict(rank=rank, probe_step=steps, UNEXPLAINED='gradient differs from the first step and no GroupNorm-backward launch '
                                      'disagreed with its repeats')), flush=True)
    q.put(dict(rank=rank, steps=steps, triple_launch_events=events, steps_with_a_different_gradient=grad_diff))


if __name__ == '__main__':
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    nproc = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    mp.set_start_method('spawn')
    q = mp.Queue()
    ps = [mp.Process(target=worker, args=(r, seconds, q)) for r in range(nproc)]
    for p in ps:
        p.start()
    res = [q.get(timeout=seconds + 600) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    print(json.dumps(dict(processes=nproc, seconds=seconds, env={k: os.environ.get(k) for k in ('HSA_ENABLE_SDMA', 'GPU_MAX_HW_QUEUES')},
                          per_process=sorted(res, key=lambda r: r['rank']))), flush=True)
