#!/usr/bin/env python
"""Training-step timing (BASELINE config #4: CO3D 10-category finetune, README.md:250-264 — seq 10, n_loss_skip 1,
global batch 80 = 10 scenes per GPU, localization weight 5, pose multiplier 0.05, dropout 0.1 = the reference default).

  python tools/bench_train.py [--steps K] [--batch 10]
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py   # DP, RCCL all-reduce
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
    ap.add_argument('--batch', type=int, default=10)
    ap.add_argument('--seq', type=int, default=10)
    ap.add_argument('--dropout', type=float, default=0.1, help='the reference default, 0.1 (counter-based masks at all four sites)')
    ap.add_argument('--precision', choices=['f32', 'bf16'], default='f32',
                    help='bf16: dense GEMMs of the forward and backward pass on bf16 MFMA (fp32 master weights, fp32 attention / '
                         'normalisation / losses / optimizer), like the reference\'s --fp16')
    ap.add_argument('--serial-wgrad', action='store_true', help='weight-gradient GEMMs on the compute stream (A/B of MIGTTrainer.overlap_weight_gradients)')
    args = ap.parse_args()
    from viewformer_amd import sharding
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    from viewformer_amd import geometry
    rank, local, world = sharding.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    cfg = MIGTConfig(sequence_size=args.seq, n_loss_skip=1, localization_weight='5', pose_multiplier=0.05, dropout=args.dropout,
                     learning_rate=1e-4, weight_decay=0.05, total_steps=40000, batch_size=80)
    model = MIGT(cfg, precision=args.precision).load_state_dict(make_migt_weights(cfg, seed=0)).to(dev)
    tr = MIGTTrainer(model)
    tr.overlap_weight_gradients = not args.serial_wgrad
    g = np.random.Generator(np.random.PCG64(rank))
    B, S = args.batch, args.seq
    tokens = torch.from_numpy(g.integers(0, 1024, size=(B, S, 8, 8))).to(dev)
    _, cams = synthetic_scene_batch(B, S, 8, seed=rank)
    poses = geometry.normalize_cameras(geometry.to_relative_cameras(torch.from_numpy(cams))[0]).to(dev)
    for _ in range(args.warmup):
        met = tr.train_step(poses, tokens)
    torch.cuda.synchronize()
    sharding.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        met = tr.train_step(poses, tokens)
    torch.cuda.synchronize()
    sharding.barrier()
    dt = sharding.max_over_ranks(time.perf_counter() - t0, dev) / args.steps
    if rank == 0:
        # forward ~0.37 TFLOP per sample (3 streams x 640 tokens), backward 2x
        tf = 3 * 0.37 * B * world
        print(json.dumps({'metric': 'MIGT training step (fwd+bwd+AdamW), CO3D-10cat config', 'ms_per_step': round(dt * 1e3, 1),
                          'samples_per_s': round(B * world / dt, 2), 'n_gpus': world, 'scenes_per_gpu': B,
                          'approx_tflops': round(tf / dt, 1), 'loss': float(met['loss']), 'dtype': args.precision, 'dropout': args.dropout, 'wgrad_stream': 'compute' if args.serial_wgrad else 'second',
                          'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
