"""CPU: the N>1 path — scene sharding + bench bookkeeping over torch.distributed (gloo, world 2)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from viewformer_amd.sharding import scene_shard


def test_scene_shard_partitions_exactly():
    for n in (0, 1, 7, 8, 64, 1000):
        for world in (1, 2, 3, 8):
            spans = [scene_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        scene_shard(4, 2, 2)


def _worker(rank, world, port, n_scenes, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from viewformer_amd import sharding
    r, _, w = sharding.init_from_env('gloo')
    a, b = sharding.scene_shard(n_scenes, r, w)
    # stand-in for "process my scenes": a deterministic per-scene value
    mine = torch.arange(a, b, dtype=torch.int64) * 3 + 1
    sharding.barrier()
    t = sharding.max_over_ranks(0.5 + r)            # max-over-ranks timing
    total = sharding.sum_over_ranks(float(b - a))
    parts = sharding.gather_to_rank0(mine)
    if r == 0:
        torch.save(dict(t=t, total=total, all=torch.cat(parts)), out)
    dist.destroy_process_group()


def test_world2_gloo_shards_cover_all_scenes(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(2, port, 7, out), nprocs=2, join=True)
    res = torch.load(out)
    assert res['t'] == 1.5 and res['total'] == 7.0
    assert torch.equal(res['all'], torch.arange(7) * 3 + 1)


def _ema_worker(rank, world, port, out):
    """two replicas, each with half of the batch: the quantizer's two all-reduces (utils_th.py:50-52) make both hold the state a
    single replica would reach on the whole batch"""
    import numpy as np
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import vqgan_oracle as vq
    from viewformer_amd import sharding
    sharding.init_from_env('gloo')
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vq_ema.npz'))
    D, K = g['E0'].shape
    state = dict(embeddings=torch.from_numpy(g['E0']), ema_cluster_size_hidden=torch.zeros(K), ema_dw_hidden=torch.zeros(D, K), counter=0)
    z = torch.from_numpy(np.concatenate([g['z0'], g['z1']], 0))          # 6 images: 3 per replica
    mine = z[rank * 3:(rank + 1) * 3]
    vq.quantize_train_step(state, mine, float(g['decay']), float(g['eps']), all_reduce=lambda t: dist.all_reduce(t))
    torch.save(state, out + f'.{rank}')
    dist.destroy_process_group()


def test_world2_quantizer_ema_allreduce_protocol(tmp_path):
    import numpy as np
    from oracle import vqgan_oracle as vq
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'ema')
    mp.spawn(_ema_worker, args=(2, port, out), nprocs=2, join=True)
    a, b = torch.load(out + '.0'), torch.load(out + '.1')
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vq_ema.npz'))
    D, K = g['E0'].shape
    one = dict(embeddings=torch.from_numpy(g['E0']), ema_cluster_size_hidden=torch.zeros(K), ema_dw_hidden=torch.zeros(D, K), counter=0)
    vq.quantize_train_step(one, torch.from_numpy(np.concatenate([g['z0'], g['z1']], 0)), float(g['decay']), float(g['eps']))
    for key in ('embeddings', 'ema_cluster_size_hidden', 'ema_dw_hidden'):
        assert torch.equal(a[key], b[key])                                # replicas agree bit for bit
        assert torch.allclose(a[key], one[key], rtol=1e-5, atol=1e-7)     # and equal the single-replica update on the whole batch
