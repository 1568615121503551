#!/usr/bin/env python
"""Codebook (VQGAN) training-step timing at the reference's codebook config (README.md codebook training: 128x128 images,
ch 128, ch_mult [1,1,2,2,4], 2 res blocks, attention at 16x16, global batch 352 = 44 per GPU on 8 GPUs, L1 + LPIPS-VGG + commitment
loss, Adam; random weights).

  python tools/bench_vqtrain.py [--steps K] [--batch 44]
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_vqtrain.py   # DP, RCCL all-reduce
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=44)
    ap.add_argument('--perceptual-weight', type=float, default=1.0, help='reference default 1.0 (LPIPS-VGG, random weights here)')
    args = ap.parse_args()
    from viewformer_amd import sharding
    from viewformer_amd.config import VQGANConfig
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.vqgan_train import VQGANTrainer
    from viewformer_amd.weights import make_vqgan_weights
    rank, local, world = sharding.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    cfg = VQGANConfig(perceptual_weight=args.perceptual_weight)
    model = VQGAN(cfg, device=dev)
    model.load_state_dict(make_vqgan_weights(cfg, seed=0))
    from viewformer_amd.lpips import make_lpips_weights
    tr = VQGANTrainer(model, lpips_state_dict=make_lpips_weights(0) if args.perceptual_weight > 0 else None)
    g = np.random.Generator(np.random.PCG64(rank))
    x = torch.from_numpy(g.uniform(-1, 1, size=(args.batch, 3, cfg.image_size, cfg.image_size)).astype(np.float32)).to(dev)
    for _ in range(args.warmup):
        met = tr.train_step(x)
    torch.cuda.synchronize()
    sharding.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        met = tr.train_step(x)
    torch.cuda.synchronize()
    sharding.barrier()
    dt = sharding.max_over_ranks(time.perf_counter() - t0, dev) / args.steps
    if rank == 0:
        print(json.dumps(dict(metric='codebook_train_images_per_sec', value=args.batch * world / dt, ms_per_step=dt * 1e3, n_gpus=world,
                              batch_per_gpu=args.batch, loss=float(met['total_loss']), rec_loss=float(met['rec_loss']), p_loss=float(met['p_loss']),
                              perceptual_weight=args.perceptual_weight,
                              peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30, dtype='f32 (split-bf16 x6)', data='synthetic')))


if __name__ == '__main__':
    main()
