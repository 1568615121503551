"""GPU: the PRODUCT trainers with more than one rank (the world-2 CPU tests only covered a 3-line helper and the
oracle).  Two processes run ``MIGTTrainer`` (overlapped per-layer SUM all-reduce, viewformer/models/migt.py:471-476,488),
``VQGANTrainer`` (DDP mean) and ``QuantizeEMATrainer`` (the two EMA all-reduces, viewformer/models/utils_th.py:50-52) on different data.
With >= 2 GPUs: one GPU per rank over RCCL (backend "nccl"); on a 1-GPU box both ranks share cuda:0 and the collectives go through gloo —
same product code path (torch.distributed all_reduce on device tensors, async handles included), only the transport differs.
Checked: reduced gradients == the sum (mean) of the ranks' local gradients, == the single-process gradient on the concatenated batch
(x world for the transformer: per-replica MEAN losses are SUMmed), replicas stay bit-identical through optimizer steps."""
import os
import socket

import numpy as np
import pytest
import torch

from conftest import TINY_MIGT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    import torch.distributed as dist
    ngpu = torch.cuda.device_count()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda', rank if ngpu >= world else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl' if ngpu >= world else 'gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    return dev, dist


def _gather(t, dist, world):
    outs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(outs, t.contiguous())
    return outs


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


def _migt_worker(rank, world, port, q):
    try:
        dev, dist = _init(rank, world, port)
        from oracle import migt_oracle as mg
        from viewformer_amd.config import MIGTConfig
        from viewformer_amd.migt import MIGT
        from viewformer_amd.train import MIGTTrainer
        from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
        cfg = MIGTConfig(**TINY_MIGT, dropout=0.0, n_loss_skip=1, localization_weight='2', pose_multiplier=0.2, learning_rate=1e-3,
                         weight_decay=0.05, total_steps=50)
        sd = make_migt_weights(cfg, seed=1, std=0.08)
        B, S, t = 2, 4, cfg.token_image_size

        def batch(seed, n=B):
            g = np.random.Generator(np.random.PCG64(seed))
            tok = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(n, S, t, t)))
            _, cams = synthetic_scene_batch(n, S, 8, seed)
            return mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(cams))[0]), tok
        tr = MIGTTrainer(MIGT(cfg).load_state_dict(sd).to(dev))
        poses, tok = batch(100 + rank)
        tr.train_step(poses, tok, reduce_gradients=False, apply_update=False)
        g_local = tr.flat_g.clone()
        tr.train_step(poses, tok, reduce_gradients=True, apply_update=False)            # per-layer ranges, async, overlapped
        g_red = tr.flat_g.clone()
        locs = _gather(g_local, dist, world)
        e_sum = _rel(g_red, sum(locs))
        # single process on the concatenated batch: gradient of the mean over 2B scenes = (g0 + g1) / world
        p_all = torch.cat([batch(100 + r)[0] for r in range(world)], 0)
        t_all = torch.cat([batch(100 + r)[1] for r in range(world)], 0)
        tr.train_step(p_all, t_all, reduce_gradients=False, apply_update=False)
        e_cat = _rel(g_red, tr.flat_g * world)
        # optimizer steps: replicas stay bit-identical
        p_before = tr.flat_p.clone()
        for s in range(2):
            tr.train_step(*batch(200 + 10 * s + rank))
        params = _gather(tr.flat_p, dist, world)
        in_sync = all(torch.equal(params[0], p) for p in params[1:])
        moved = float((tr.flat_p - p_before).abs().max())
        q.put((rank, 'ok', dict(e_sum=e_sum, e_cat=e_cat, in_sync=in_sync, moved=moved, backend=dist.get_backend())))
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, 'err', traceback.format_exc() + repr(e)))


def _vq_worker(rank, world, port, q):
    try:
        dev, dist = _init(rank, world, port)
        sharing = torch.cuda.device_count() < world          # ranks time-sliced on one GPU: the triple-launch detector rides along
        from viewformer_amd.config import VQGANConfig
        from viewformer_amd.vq_train import QuantizeEMATrainer
        from viewformer_amd.vqgan import VQGAN
        from viewformer_amd.vqgan_train import VQGANTrainer
        from viewformer_amd.weights import make_vqgan_weights
        cfg = VQGANConfig(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], image_size=32, z_channels=32, embed_dim=32,
                          n_embed=64, perceptual_weight=0.0, codebook_weight=1.0, learning_rate=1e-3)
        sd = make_vqgan_weights(cfg, seed=3, codebook_scale=0.05)

        def images(seed, n=3):
            g = np.random.Generator(np.random.PCG64(seed))
            return torch.from_numpy((g.random((n, 3, 32, 32)) * 2 - 1).astype(np.float32))
        # ---- VQGANTrainer: Lightning DDP = mean of the replicas' gradients (vqgan_th.py:415-422 under accelerator='ddp')
        model = VQGAN(cfg, device=dev)
        model.load_state_dict(sd)
        tr = VQGANTrainer(model)
        tr.debug_triple_groupnorm_bwd = sharing
        state0 = {k: v.clone() for k, v in tr.quantizer_state().items()} if hasattr(tr, 'quantizer_state') else None
        tr.train_step(images(50 + rank), reduce_gradients=False, apply_update=False)
        g_local = tr.flat_g.clone()
        locs = _gather(g_local, dist, world)
        # a fresh trainer for the reduced step: the EMA codebook moved during the first forward
        model2 = VQGAN(cfg, device=dev)
        model2.load_state_dict(sd)
        tr2 = VQGANTrainer(model2)
        tr2.debug_triple_groupnorm_bwd = sharing
        tr2.train_step(images(50 + rank), reduce_gradients=True, apply_update=False)
        e_mean = _rel(tr2.flat_g, sum(locs) / world)
        for s in range(2):
            tr2.train_step(images(70 + 10 * s + rank))
        params = _gather(tr2.flat_p, dist, world)
        in_sync = all(torch.equal(params[0], p) for p in params[1:])
        # ---- QuantizeEMATrainer: counts / embed_sum SUMmed over replicas (utils_th.py:50-52) == one process on the concatenated rows
        E = torch.from_numpy(np.asarray(sd['quantize.embeddings'])).to(dev)

        def zbatch(seed):
            g = np.random.Generator(np.random.PCG64(seed))
            return torch.from_numpy((g.standard_normal((2, 32, 4, 4)) * 0.1).astype(np.float32)).to(dev)
        qt = QuantizeEMATrainer(E)
        for s in range(2):
            qt(zbatch(300 + 10 * s + rank))
        Es = _gather(qt.embeddings, dist, world)
        q_sync = all(torch.equal(Es[0], e) for e in Es[1:])
        # the single-process reference needs no collective: build it with a one-rank group
        solo = [dist.new_group([r]) for r in range(world)][rank]          # (new_group is collective: every rank creates every group)
        ref = QuantizeEMATrainer(E, process_group=solo)
        for s in range(2):
            ref(torch.cat([zbatch(300 + 10 * s + r) for r in range(world)], 0))
        e_ema = _rel(qt.embeddings, ref.embeddings)
        q.put((rank, 'ok', dict(e_mean=e_mean, in_sync=in_sync, q_sync=q_sync, e_ema=e_ema, backend=dist.get_backend(),
                                transient_events=tr.transient_events + tr2.transient_events)))
        del state0
        dist.destroy_process_group()
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, 'err', traceback.format_exc() + repr(e)))


def _run(worker, world=2):
    import torch.multiprocessing as mp
    assert torch.cuda.is_available(), 'needs the MI355X'
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, status, payload = q.get(timeout=600)
        assert status == 'ok', f'rank {rank}: {payload}'
        res[rank] = payload
    for p in procs:
        p.join(timeout=60)
    return res


def _report(msg):
    print('TRANSIENT', msg)
    try:
        import json
        from conftest import REPO
        os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(REPO, 'gpurun_out', 'parity_report.jsonl'), 'a') as f:
            f.write(json.dumps(msg, default=str) + '\n')
    except OSError:
        pass


def _run_proving_the_known_transient(worker, ok):
    """No blind retry.  Two ranks on ONE GPU is a test-only configuration in which this code base has shown
    exactly one transient: with several processes time-sliced on the device, one of three identical GroupNorm-backward launches returns lanes
    48-63 of one accumulator wrong (profiles/r5_gpu_sharing_transient.txt, r6_gpu_sharing_transient.txt).  The codebook-trainer workers
    therefore run with ``VQGANTrainer.debug_triple_groupnorm_bwd`` on whenever the ranks share a device: every GroupNorm backward is issued
    three times and compared on the spot, launch #0's result is used as the product would use it, and events travel back with the result.
    A missed bound is FATAL unless that very run recorded an event whose outlier was launch #0 (the one the step consumed) — i.e. unless the
    run itself proved that the known transient, and nothing else, hit it.  Only then is the run repeated, once, and must pass.  With one GPU
    per rank (RCCL) the detector is off and nothing is ever repeated."""
    res = _run(worker)
    if ok(res):
        evs = [e for m in res.values() for e in m.get('transient_events', [])]
        if evs:
            _report(dict(test=worker.__name__, note='bounds met; GroupNorm-backward repeats disagreed in launches the step did not consume', events=evs))
        return res
    consumed = [e for m in res.values() for e in m.get('transient_events', []) if e.get('outlier_launch') == 0]
    assert torch.cuda.device_count() < 2 and consumed, \
        f'bound missed and the run recorded no consumed triple-launch outlier: not the known GPU-sharing transient -> a real failure: {res}'
    _report(dict(test=worker.__name__, note='bound missed; the same run caught the known transient in a launch the step consumed; repeated once',
                 first=res))
    return _run(worker)


def test_migt_trainer_world2_sum_allreduce():
    res = _run(_migt_worker)                     # never repeated: no known transient touches the transformer trainer
    print(res)
    for r, m in res.items():
        assert m['e_sum'] < 1e-6, m             # overlapped per-layer all-reduce == sum of the ranks' gradients
        assert m['e_cat'] < 1e-4, m             # == world x the single-process gradient on the concatenated batch (mean-loss, SUM)
        assert m['in_sync'] and m['moved'] > 0, m


def test_codebook_trainers_world2_mean_and_ema_allreduce():
    res = _run_proving_the_known_transient(_vq_worker, lambda r: all(m['e_mean'] < 1e-6 and m['e_ema'] < 1e-5 for m in r.values()))
    print(res)
    for r, m in res.items():
        assert m['e_mean'] < 1e-6, m            # DDP mean of the replicas' gradients
        assert m['in_sync'] and m['q_sync'], m  # replicas bit-identical after optimizer / EMA steps
        assert m['e_ema'] < 1e-5, m             # EMA codebook == one process on the concatenated rows
