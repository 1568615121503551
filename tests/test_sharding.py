"""CPU: the N>1 path — scene sharding + bench bookkeeping over torch.distributed (gloo, world 2 and world 8)."""
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


def _world8_worker(rank, world, port, n_scenes, out):
    """what each of the eight ranks of the driver's 8-GPU line does, minus the kernels: its scene shard, the bookkeeping collectives, the
    trainer's per-layer-range gradient SUM (no division: migt.py:471-476 + MirroredStrategy, train/utils.py:145-153)"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from viewformer_amd import sharding
    r, _, w = sharding.init_from_env('gloo')
    a, b = sharding.scene_shard(n_scenes, r, w)
    mine = torch.arange(a, b, dtype=torch.int64) * 5 + 2
    sharding.barrier()
    t = sharding.max_over_ranks(1.0 + 0.25 * r)
    total = sharding.sum_over_ranks(float(b - a))
    parts = sharding.gather_to_rank0(mine)
    # the flat gradient buffer: every rank holds a different integer-valued gradient (sums are exact in fp32), ranges like the trainer's
    # (uneven, one empty, one single element); elements outside every range must stay local
    g = torch.Generator().manual_seed(100 + r)
    flat = torch.randint(-8, 9, (10_007,), generator=g).float()
    local = flat.clone()
    ranges = [(0, 4096), (4096, 4096), (4096, 4097), (4097, 9000)]
    for h in sharding.allreduce_sum_ranges(flat, ranges):
        h.wait()
    torch.save(dict(t=t, total=total, all=torch.cat(parts) if r == 0 else None, flat=flat, local=local), out + f'.{r}')
    dist.destroy_process_group()


def test_world8_gloo_scene_shards_and_gradient_ranges(tmp_path):
    """the 8-rank form of everything the multi-GPU lines lean on (BASELINE configs[3] / [4] name 8 x MI355X): 61 scenes over 8 ranks
    (uneven shards), max / sum bookkeeping, ragged gather, per-range SUM all-reduce of the flat gradient buffer"""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'w8')
    mp.spawn(_world8_worker, args=(8, port, 61, out), nprocs=8, join=True)
    res = [torch.load(out + f'.{r}') for r in range(8)]
    assert all(r['t'] == 2.75 and r['total'] == 61.0 for r in res)
    assert torch.equal(res[0]['all'], torch.arange(61) * 5 + 2)
    want = sum(r['local'] for r in res)
    for r in res:
        assert torch.equal(r['flat'][:9000], want[:9000])                # SUM over the 8 replicas, not a mean
        assert torch.equal(r['flat'][9000:], r['local'][9000:])          # outside the ranges: untouched


def _ema8_worker(rank, world, port, out):
    import numpy as np
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import vqgan_oracle as vq
    from viewformer_amd import sharding
    sharding.init_from_env('gloo')
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vq_ema.npz'))
    D, K = g['E0'].shape
    state = dict(embeddings=torch.from_numpy(g['E0']), ema_cluster_size_hidden=torch.zeros(K), ema_dw_hidden=torch.zeros(D, K), counter=0)
    z = torch.from_numpy(np.concatenate([g['z0'], g['z1']] * 4, 0))          # 24 images: 3 per replica
    vq.quantize_train_step(state, z[rank * 3:(rank + 1) * 3], float(g['decay']), float(g['eps']), all_reduce=lambda t: dist.all_reduce(t))
    if rank in (0, 7):
        torch.save(state, out + f'.{rank}')
    dist.destroy_process_group()


def test_world8_quantizer_ema_allreduce_protocol(tmp_path):
    """eight replicas with three images each == one replica on the 24 images (utils_th.py:46-64: the two EMA all-reduces)"""
    import numpy as np
    from oracle import vqgan_oracle as vq
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'ema8')
    mp.spawn(_ema8_worker, args=(8, port, out), nprocs=8, join=True)
    a, b = torch.load(out + '.0'), torch.load(out + '.7')
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vq_ema.npz'))
    D, K = g['E0'].shape
    one = dict(embeddings=torch.from_numpy(g['E0']), ema_cluster_size_hidden=torch.zeros(K), ema_dw_hidden=torch.zeros(D, K), counter=0)
    vq.quantize_train_step(one, torch.from_numpy(np.concatenate([g['z0'], g['z1']] * 4, 0)), float(g['decay']), float(g['eps']))
    for key in ('embeddings', 'ema_cluster_size_hidden', 'ema_dw_hidden'):
        assert torch.equal(a[key], b[key])
        assert torch.allclose(a[key], one[key], rtol=1e-5, atol=1e-7)


def test_bare_bench_refuses_more_ranks_than_gpus():
    """``python bench.py --gpus N`` with no launcher must never print a line for fewer ranks than it was asked for (it
    used to run ONE rank and report n_gpus 1).  Here there is no GPU at all: --gpus 2 has to exit non-zero with a one-line reason and no
    JSON line; a launcher whose WORLD_SIZE disagrees with --gpus is refused the same way."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'VF_DIST_BACKEND')}
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip('box has >= 2 GPUs: the bare command is allowed to run here')
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], cwd=repo, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'refusing' in r.stderr and '--gpus 2' in r.stderr, r.stderr[-500:]
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '4', '--steps', '1', '--warmup', '0'], cwd=repo,
                       env=dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_bare_bench_with_the_gloo_switch_reaches_the_launcher():
    """the other branch of bench.py's self-launch on a box without GPUs: with VF_DIST_BACKEND=gloo the bare ``--gpus 2`` is NOT refused — it starts
    ``torch.distributed.run`` with two ranks (announced on stderr), each of which then fails loudly because the hot path has no CPU fallback; the
    parent returns the launcher's non-zero status and prints no JSON line.  (On a GPU box the same command produces the n_gpus = 2 line:
    tests/test_hip_multigpu.py.)"""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip('CPU-box branch')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['VF_DIST_BACKEND'] = 'gloo'
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--batch', '1'], cwd=repo, env=env,
                       capture_output=True, text=True, timeout=600)
    assert 'starting 2 ranks' in r.stderr and '--nproc-per-node=2' in r.stderr, r.stderr[-800:]
    assert r.returncode != 0 and 'bench.py needs the MI355X' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
