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
