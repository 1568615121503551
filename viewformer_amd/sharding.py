"""Scene sharding for multi-GPU inference.

Scenes are independent (``generate_batch_predictions`` keeps no cross-batch state,
evaluate_transformer.py:97-146,219), so N GPUs = N processes each owning a contiguous range
of scenes with replicated weights and NO data-path collective.  The only communication is the
bench/evaluator bookkeeping below (barrier, max-over-ranks time, optional gather of results),
over torch.distributed (RCCL on GPUs, gloo in the CPU tests).
"""
import os

import torch
import torch.distributed as dist


def scene_shard(n_scenes: int, rank: int, world: int):
    """contiguous, balanced [start, stop) of scenes owned by ``rank``"""
    if world <= 0 or not (0 <= rank < world) or n_scenes < 0:
        raise ValueError('bad shard arguments')
    base, rem = divmod(n_scenes, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_from_env(backend: str = None):
    """one process per GPU, launched by torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)"""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # VF_DIST_BACKEND=gloo: the same N-process code path on a box with fewer GPUs than ranks (ranks share devices, the collectives carry
    # device tensors over gloo) — how the 1-GPU test boxes exercise bench.py's N > 1 branches; the product launch leaves it unset (RCCL)
    backend = backend or os.environ.get('VF_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        if backend == 'gloo':
            local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)                      # bind this rank's GPU BEFORE RCCL creates its communicator
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC (the host driver has no legacy IPC)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device='cpu') -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device='cpu') -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def allreduce_sum_ranges(flat: torch.Tensor, ranges, group=None):
    """The training step's one collective (the 88.4 M-element gradient all-reduce): SUM over
    replicas of contiguous ranges of the flat gradient buffer — no division by the world size, because the
    reference sums per-replica mean-loss gradients (migt.py:471-476 + MirroredStrategy).  One asynchronous
    RCCL (gloo in the CPU tests) all-reduce per range (a transformer layer = 28 MB fp32, large enough to run at
    xGMI link rate); returns the work handles."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return []
    return [dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=group, async_op=True) for a, b in ranges if b > a]


def gather_to_rank0(t: torch.Tensor):
    """optional: collect per-rank result tensors (e.g. uint8 novel views) on rank 0"""
    if not (dist.is_available() and dist.is_initialized()):
        return [t]
    world = dist.get_world_size()
    sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device))
    mx = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    outs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return [o[:int(s.item())] for o, s in zip(outs, sizes)]
