"""Training step.  CPU: oracle schedule / optimizer restatement and the DP gradient
all-reduce over gloo (world 2).  GPU: every backward kernel against torch autograd, the full-step gradients
against fp64 autograd over the oracle, and the AdamWeightDecay update against its numpy restatement."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from conftest import TINY_MIGT


# ------------------------------------------------------------------------------------------------ CPU
def test_schedules_match_reference_formulas():
    from oracle import train_oracle as to
    from viewformer_amd.train import learning_rate, parse_schedule
    # WarmUp: lr * step/warmup below warmup; CosineDecay(init, total - warmup) after (models/utils.py:346-361,403-412)
    assert learning_rate(0, 1e-4, 40000, 2000) == 0.0
    assert abs(learning_rate(1000, 1e-4, 40000, 2000) - 0.5e-4) < 1e-12
    assert abs(learning_rate(2000, 1e-4, 40000, 2000) - 1e-4) < 1e-12
    assert abs(learning_rate(21000, 1e-4, 40000, 2000) - 0.5e-4) < 1e-12
    assert learning_rate(40000, 1e-4, 40000, 2000) < 1e-12 and learning_rate(10 ** 6, 1e-4, 40000, 2000) < 1e-12
    for s in (0, 7, 1999, 2000, 12345, 39999):
        assert abs(learning_rate(s, 6.4e-4, 40000, 2000) - to.learning_rate(s, 6.4e-4, 40000, 2000)) < 1e-15
    f = parse_schedule('cosine(0,1,120000)')                      # README.md:356 (SM7)
    assert f(0) == 0.0 and abs(f(60000) - 0.5) < 1e-12 and abs(f(120000) - 1.0) < 1e-12 and f(10 ** 7) == 1.0
    assert parse_schedule('5.')(123) == 5.0
    assert abs(to.schedule_value('cosine(0,1,120000)', 30000) - f(30000)) < 1e-15


def test_learning_rate_offset_of_the_finetune_driver():
    """WarmUp.offset (models/utils.py:337,341; finetune_transformer.py:86): the schedule counts from the restored iteration count"""
    from oracle import train_oracle as to
    from viewformer_amd.train import learning_rate
    assert learning_rate(120000, 1e-5, 10000, 2000, offset=120000) == 0.0
    assert abs(learning_rate(121000, 1e-5, 10000, 2000, offset=120000) - 0.5e-5) < 1e-18
    assert learning_rate(100, 1e-5, 10000, 2000, offset=120000) == 0.0                   # max(step - offset, 0)
    for s_ in (120000, 120007, 123456, 131000):
        assert learning_rate(s_, 1e-5, 10000, 2000, 120000) == to.learning_rate(s_, 1e-5, 10000, 2000, 120000) == learning_rate(s_ - 120000, 1e-5, 10000, 2000)


@pytest.mark.parametrize('augment', ['relative', 'no', 'simple', 'advanced'])
def test_process_batch_pose_augmentation(augment):
    """the token dataset's per-sequence transform (train_transformer.py:28-61) against the fp64 restatement, with the random numbers handed to both"""
    import math
    from oracle import train_oracle as to
    from oracle import migt_oracle as mg
    from viewformer_amd.train import process_batch, pose_augmentation_draws
    from viewformer_amd.weights import synthetic_scene_batch
    B, S = 3, 5
    _, cams = synthetic_scene_batch(B, S, 8, 11)
    cams = torch.from_numpy(cams)
    cams[1, :, 3:] *= -1.0                                       # a sign the sign-fix has to undo
    tokens = torch.arange(B * S).view(B, S)
    gen = torch.Generator().manual_seed(5)
    draws = pose_augmentation_draws(augment, B, gen)
    out, tok = process_batch(cams, tokens, augment, 'train', draws=draws)
    assert tok is tokens and out.shape == cams.shape and out.dtype == cams.dtype
    for b in range(B):
        d = {k: (v[b].double().numpy() if v[b].dim() else float(v[b])) for k, v in draws.items()}
        ref = to.process_batch_np(cams[b].numpy(), augment, 'train', d)
        assert np.abs(out[b].double().numpy() - ref).max() < 5e-6, (augment, b)
        one, _ = process_batch(cams[b], tokens[b], augment, 'train', draws={k: v[b:b + 1] for k, v in draws.items()})   # [S,7] form
        assert torch.equal(one, out[b])
    assert (out[..., 3] >= 0).all() and ((out[..., 3:] ** 2).sum(-1) - 1).abs().max() < 1e-5
    if augment == 'relative':                                    # the evaluators' frame change is the same map (evaluate_transformer.py:70-78)
        assert torch.allclose(out, mg.normalize_cameras(mg.to_relative_cameras(cams)[0]), atol=1e-6)
        assert out[:, 0, :3].abs().max() < 1e-6 and (out[:, 0, 3:] - torch.tensor([1.0, 0, 0, 0])).abs().max() < 1e-6
    if augment in ('simple', 'advanced'):
        same, _ = process_batch(cams, tokens, augment, 'test')   # outside the training split nothing is drawn (:37-38)
        base, _ = process_batch(cams, tokens, 'no', 'train')
        assert torch.equal(same, base) and not torch.allclose(out, base)
        assert set(draws) == ({'shift', 'y0', 'x', 'y1'} if augment == 'simple' else {'shift', 'y0'})
        assert float(draws['y0'].max()) < 2 * math.pi and ('x' not in draws or float(draws['x'].max()) < math.pi / 8)
        a, _ = process_batch(cams, tokens, augment, 'train', generator=torch.Generator().manual_seed(9))
        b_, _ = process_batch(cams, tokens, augment, 'train', generator=torch.Generator().manual_seed(9))
        assert torch.equal(a, b_)
        # a rigid change of the world frame: distances between the views' positions are untouched
        assert torch.allclose(torch.cdist(out[..., :3], out[..., :3]), torch.cdist(cams[..., :3], cams[..., :3]), atol=1e-4)
    with pytest.raises(ValueError):
        process_batch(cams, tokens, 'fancy', 'train')


def test_oracle_adam_weight_decay_step():
    from oracle import train_oracle as to
    from viewformer_amd.config import MIGTConfig
    cfg = MIGTConfig(learning_rate=1e-3, weight_decay=0.05, total_steps=100)
    p = {'w.weight': np.array([1.0, -2.0]), 'w.bias': np.array([0.5])}
    g = {'w.weight': np.array([0.1, 0.2]), 'w.bias': np.array([-0.3])}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(v_) for k, v_ in p.items()}
    lr = to.adam_weight_decay_step(p, g, m, v, step=5, cfg=cfg, warmup_steps=10)
    assert abs(lr - 0.5e-3) < 1e-15
    lr_t = lr * math.sqrt(1 - 0.999 ** 6) / (1 - 0.9 ** 6)
    # weight: decayed then Adam; bias: Adam only
    w0 = 1.0 - lr * 1.0 * 0.05
    assert abs(p['w.weight'][0] - (w0 - lr_t * (0.1 * 0.1) / (math.sqrt(0.001 * 0.01) + 1e-8))) < 1e-12
    assert abs(p['w.bias'][0] - (0.5 - lr_t * (0.1 * -0.3) / (math.sqrt(0.001 * 0.09) + 1e-8))) < 1e-12


def _ar_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from viewformer_amd import sharding
    sharding.init_from_env('gloo')
    flat = torch.arange(40, dtype=torch.float32) * (rank + 1)
    handles = sharding.allreduce_sum_ranges(flat, [(0, 8), (8, 8), (8, 40)])
    for h in handles:
        h.wait()
    if rank == 0:
        torch.save(flat, out)
    dist.destroy_process_group()


def test_gradient_allreduce_is_a_sum_over_replicas(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'g.pt')
    mp.spawn(_ar_worker, args=(2, port, out), nprocs=2, join=True)
    assert torch.equal(torch.load(out), torch.arange(40, dtype=torch.float32) * 3)      # SUM, not mean


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs the MI355X'
    return torch.device('cuda:0')


def _rand(shape, seed, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def _err(a, b):
    b = torch.as_tensor(b).double()
    return ((a.detach().cpu().double() - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.gpu
def test_backward_kernels_against_autograd(dev):
    from viewformer_amd import train_ops as T
    # transpose (ragged, batched)
    x = _rand((3, 70, 45), 1)
    t = T.transpose(x.to(dev), 70, 45, batch=3, bs_src=70 * 45)
    assert torch.equal(t.cpu(), x.transpose(1, 2).contiguous())
    # column sums
    y = _rand((1000, 130), 2)
    out = torch.ones(130, device=dev)
    T.colsum(y.to(dev), out, 1000, 130, accumulate=True)
    assert _err(out, 1 + y.double().sum(0)) < 1e-5
    # LayerNorm backward
    rows, d = 150, 768
    xx, dy, g = (_rand((rows, d), 3) * 2 + 0.3).double().requires_grad_(), _rand((rows, d), 4).double(), (_rand((d,), 5) + 1).double().requires_grad_()
    b = torch.zeros(d, dtype=torch.float64, requires_grad=True)
    F.layer_norm(xx, (d,), g, b, eps=1e-5).backward(dy)
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    dx = T.layernorm_bwd(dy.float().to(dev), xx.detach().float().to(dev), g.detach().float().to(dev), dg, db, rows, d)
    assert _err(dx, xx.grad) < 2e-5 and _err(dg, g.grad) < 2e-5 and _err(db, b.grad) < 2e-5
    # GELU fwd/bwd
    u = (_rand((4000,), 6) * 2).double().requires_grad_()
    F.gelu(u).backward(torch.ones_like(u) * 0.7)
    assert _err(T.gelu(u.detach().float().to(dev)), F.gelu(u.detach())) < 1e-6
    assert _err(T.gelu_bwd(u.detach().float().to(dev), torch.full((4000,), 0.7, device=dev)), u.grad) < 1e-5
    # masked softmax fwd/bwd (streams mask, Sv = 2 views per stream, 3 streams, L = 16)
    B, Tn, L, Sv = 2, 96, 16, 2
    s = (_rand((B, Tn, Tn), 7) * 3).double().requires_grad_()
    qv = torch.arange(Tn) // L
    qs, qi, ks, ki = qv[:, None] // Sv, qv[:, None] % Sv, qv[None] // Sv, qv[None] % Sv
    vis = torch.where(qs == 0, (ks == 0) & (ki <= qi), ((ks == 0) & (ki < qi)) | (qv[:, None] == qv[None])).double()
    p = torch.softmax(s * vis - 1e4 * (1 - vis), -1)
    dp = _rand((B, Tn, Tn), 8).double()
    p.backward(dp)
    pg = T.softmax_mask_(s.detach().float().to(dev).clone(), B, Tn, L, -Sv)
    assert _err(pg, p.detach()) < 1e-6
    dsg = T.softmax_mask_bwd_(pg, dp.float().to(dev).clone(), B, Tn, L, -Sv)
    assert _err(dsg, s.grad) < 1e-5
    # softmax cross-entropy
    lg = (_rand((64, 1024), 9) * 2).double().requires_grad_()
    tg = torch.from_numpy(np.random.Generator(np.random.PCG64(10)).integers(0, 1024, 64))
    w = torch.linspace(0, 1, 64).double()
    (F.cross_entropy(lg, tg, reduction='none') * w).sum().backward()
    loss, dl = T.softmax_ce(lg.detach().float().to(dev), tg.int().to(dev), w.float().to(dev), 64, 1024)
    assert _err(loss, F.cross_entropy(lg.detach(), tg, reduction='none')) < 1e-6 and _err(dl, lg.grad) < 1e-5
    # AdamWeightDecay
    pw, gw = _rand((1000,), 11), _rand((1000,), 12)
    m0, v0 = _rand((1000,), 13) * 0.1, _rand((1000,), 14).abs() * 0.01
    pd, md, vd = pw.to(dev).clone(), m0.to(dev).clone(), v0.to(dev).clone()
    T.adamw_(pd, gw.to(dev), md, vd, 1e-3 * 0.05, 2e-3, 0.9, 0.999, 1e-8)
    pr = pw.double() * (1 - 1e-3 * 0.05)
    mr = 0.9 * m0.double() + 0.1 * gw.double()
    vr = 0.999 * v0.double() + 0.001 * gw.double() ** 2
    pr = pr - 2e-3 * mr / (vr.sqrt() + 1e-8)
    # (1 - beta2) is formed in fp32 like Keras does (0.00100004673 vs 0.001): 5e-5 relative on v
    assert _err(pd, pr) < 1e-5 and _err(md, mr) < 1e-6 and _err(vd, vr) < 1e-4
    # clip_by_norm
    z = _rand((5000,), 15)
    zc = T.clip_by_norm_(z.to(dev).clone(), 3.0, torch.zeros(1, device=dev))
    assert _err(zc, z.double() * 3.0 / max(z.double().norm().item(), 3.0)) < 1e-6


def _setup(loc, dev, seed=1, B=2, S=4, clip=0.0, precision='f32', label_smoothing=0.0, dense_arith='x3h', **extra):
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    from oracle import migt_oracle as mg
    cfg = MIGTConfig(**TINY_MIGT, dropout=0.0, n_loss_skip=1, localization_weight='cosine(0,2,10)' if loc else '0',
                     pose_multiplier=0.2, learning_rate=1e-3, weight_decay=0.05, total_steps=50, gradient_clip_val=clip,
                     label_smoothing=label_smoothing, **extra)
    sd = make_migt_weights(cfg, seed=seed, std=0.08)
    g = np.random.Generator(np.random.PCG64(seed + 3))
    t = cfg.token_image_size
    tokens = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(B, S, t, t)))
    _, cams = synthetic_scene_batch(B, S, 8, seed)
    poses = mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(cams))[0])       # process_batch, train_transformer.py:31-64
    model = MIGT(cfg, precision=precision, dense_arith=dense_arith).load_state_dict(sd).to(dev)
    return cfg, sd, tokens, poses, MIGTTrainer(model, warmup_steps=4)


@pytest.mark.gpu
@pytest.mark.parametrize('loc,arith', [(False, 'x3h'), (True, 'x3h'), (True, 'x6'), (True, 'f32')])
def test_train_step_gradients_match_autograd(dev, loc, arith):
    """dense_arith: x3h = forward GEMMs on the split-fp16 kernel, backward on x6 (the default); x6 / f32 = everything on that arm"""
    from oracle import train_oracle as to
    cfg, sd, tokens, poses, tr = _setup(loc, dev, dense_arith=arith)
    tr.step_count = 3                       # a non-trivial localization weight
    metrics = tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
    grads, ref_metrics = to.gradients(sd, cfg, poses, tokens, step=3)
    assert abs(float(metrics['loss']) - ref_metrics['loss']) < 1e-4 * max(1.0, abs(ref_metrics['loss']))
    assert abs(float(metrics['ce_loss']) - ref_metrics['ce_loss']) < 1e-4
    worst = ('', 0.0)
    for name in tr.names:
        ref = grads[name].reshape(tr.slices[name][2])
        got = tr.g(name)
        if not loc and name.startswith('pose_criterion'):
            assert float(got.abs().max()) == 0.0
            continue
        e = _err(got, ref)
        if e > worst[1]:
            worst = (name, e)
        assert e < 2e-3, (name, e, float(ref.abs().max()))
    print('worst relative gradient error', worst)


@pytest.mark.gpu
def test_train_step_matches_hugging_face_gpt2_autograd_golden(dev):
    """the HIP training step against a THIRD-PARTY graph: tests/golden/migt_hf_gpt2.npz holds the loss terms and, per variable, the gradient's
    norm and 48 sampled entries of MIGT.train_step's forward + losses built on ``transformers``' GPT-2 with torch autograd
    (tests/golden/make_hf_gpt2_golden.py::train_graph).  fp32-equivalent arm, dropout 0, constant localization weight."""
    from conftest import load_golden
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights
    g = load_golden('migt_hf_gpt2.npz')
    cfg = MIGTConfig(n_embeddings=64, n_head=2, d_model=128, n_layer=2, token_image_size=4, sequence_size=4, localization_weight='2', pose_multiplier=0.2,
                     n_loss_skip=1, dropout=0.0, learning_rate=1e-3, weight_decay=0.05, total_steps=50)
    sd = make_migt_weights(cfg, seed=int(g['tiny_seed']), std=float(g['tiny_std']))
    tr = MIGTTrainer(MIGT(cfg).load_state_dict(sd).to(dev), warmup_steps=4)
    met = tr.train_step(torch.from_numpy(g['tiny_cams']), torch.from_numpy(g['tiny_ids']), reduce_gradients=False, apply_update=False)
    assert abs(float(met['loss']) - float(g['train_loss'])) < 1e-4 * float(g['train_loss'])
    assert abs(float(met['ce_loss']) - float(g['train_ce'])) < 1e-4 * float(g['train_ce'])
    assert abs(float(met['pose_pos_loss']) - float(g['train_pos'])) < 1e-4 and abs(float(met['pose_ori_loss']) - float(g['train_ori'])) < 1e-4
    worst = ('', 0.0)
    for i, n in enumerate(str(x) for x in g['train_names']):
        got = tr.g(n).reshape(-1).double().cpu()
        ref_norm = float(g['train_norms'][i])
        e = max(abs(float(got.norm()) - ref_norm), float((got[torch.from_numpy(g['train_idx'][i])] - torch.from_numpy(g['train_samples'][i]).double()).abs().max()))
        if e / max(ref_norm, 1e-12) > worst[1]:
            worst = (n, e / max(ref_norm, 1e-12))
        assert e < 2e-3 * max(ref_norm, 1e-9), (n, e, ref_norm)
    print('HIP training step vs HF GPT-2 autograd golden: worst gradient deviation (of the tensor norm)', worst)


@pytest.mark.gpu
def test_train_steps_follow_the_optimizer_restatement(dev):
    """3 full steps (forward, backward, AdamWeightDecay with warm-up) vs autograd + numpy optimizer on the oracle"""
    from oracle import train_oracle as to
    cfg, sd, tokens, poses, tr = _setup(True, dev, seed=2)
    params = {k: np.asarray(v, dtype=np.float64).copy() for k, v in sd.items()}
    params = {k: (v.reshape(-1) if k.endswith('.bias') else v) for k, v in params.items()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v2 = {k: np.zeros_like(v) for k, v in params.items()}
    losses = []
    for step in range(3):
        met = tr.train_step(poses, tokens)
        grads, ref = to.gradients(params, cfg, poses, tokens, step=step)
        to.adam_weight_decay_step(params, {k: grads[k].numpy().reshape(params[k].shape) for k in params}, m, v2, step, cfg,
                                  warmup_steps=4)
        losses.append((float(met['loss']), ref['loss']))
    for a, b in losses:
        assert abs(a - b) < 2e-4 * max(1.0, abs(b)), losses
    assert losses[2][0] < losses[0][0]                         # it learns on a fixed batch
    new = tr.state_dict()
    lr_sum = sum(to.learning_rate(s, cfg.learning_rate, cfg.total_steps, 4) for s in range(3))
    for k in tr.names:
        ref = torch.as_tensor(params[k].reshape(new[k].shape))
        diff = (new[k].double() - ref).abs().max().item()
        # Adam normalises by sqrt(v): a coordinate whose true gradient is ~0 (e.g. the key bias of c_attn — softmax is
        # invariant to it) moves by +-lr on rounding noise alone, in the reference too; allow that much absolute slack.
        assert diff < 2e-3 * ref.abs().max().item() + 0.2 * lr_sum, (k, diff)


@pytest.mark.gpu
def test_optimizer_state_round_trip_and_finetune_schedule(dev):
    """resume: weights + Adam moments + iteration count restored into a fresh trainer continue bit-identically; finetune
    (finetune_transformer.py:76-86): the new schedule starts its warm-up at the restored count while Adam's bias correction goes on"""
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer, learning_rate
    cfg, sd, tokens, poses, tr = _setup(True, dev)
    for _ in range(3):
        tr.train_step(poses, tokens)
    w, o = tr.state_dict(), tr.optimizer_state_dict()
    assert o['iterations'] == 3 and o['lr_offset'] == 0 and o['m/wte.weight'].shape == w['wte.weight'].shape
    full = dict(sd)
    full.update(w)
    tr2 = MIGTTrainer(MIGT(cfg, dense_arith='x3h').load_state_dict(full).to(dev), warmup_steps=4).load_optimizer_state_dict(o)
    assert tr2.step_count == 3
    a, b = tr.train_step(poses, tokens), tr2.train_step(poses, tokens)
    assert a['loss'] == b['loss'] and torch.equal(tr.flat_p, tr2.flat_p) and torch.equal(tr.flat_m, tr2.flat_m) and torch.equal(tr.flat_v, tr2.flat_v)
    bad = dict(o)
    bad.pop('v/wte.weight')
    with pytest.raises(RuntimeError):
        MIGTTrainer(MIGT(cfg, dense_arith='x3h').load_state_dict(full).to(dev)).load_optimizer_state_dict(bad)
    # finetune: lr(iterations = offset) = 0 -> the first step moves no weight (weight decay is lr-scaled) but feeds the moments
    tr2.begin_finetune(learning_rate=1e-5, total_steps=100, warmup_steps=10)
    assert tr2.lr_offset == 4 and tr2.cfg.total_steps == cfg.total_steps
    p0, m0 = tr2.flat_p.clone(), tr2.flat_m.clone()
    tr2.train_step(poses, tokens)
    assert torch.equal(tr2.flat_p, p0) and not torch.equal(tr2.flat_m, m0) and tr2.step_count == 5
    tr2.train_step(poses, tokens)
    lr = learning_rate(5, 1e-5, 100, 10, offset=4)
    assert abs(lr - 1e-6) < 1e-18
    d = (tr2.flat_p - p0).abs().max().item()
    assert 0 < d <= 4.0 * lr * (1.0 + cfg.weight_decay * p0.abs().max().item())          # an Adam step is O(lr) per element: the finetune rate, not the 1e-3 of the run before
    assert tr2.optimizer_state_dict()['lr_offset'] == 4


@pytest.mark.gpu
def test_checkpoint_save_then_finetune_entry_point(dev, tmp_path):
    """the optimizer state reaches a checkpoint and comes back through the finetune entry point.  ``save_training_checkpoint``
    leaves what the reference's ModelCheckpoint leaves (config.json + one TF-format checkpoint of the compiled model, train/utils.py:46-86);
    ``finetune_transformer`` (finetune_transformer.py:57-86) loads it with config overrides, restores weights AND Adam state, and starts
    the finetune schedule at the restored iteration count.  The restored trainer's next step equals the original trainer's next step
    under the same schedule, bit for bit."""
    from viewformer_amd import checkpoint as ck
    from viewformer_amd.train import finetune_transformer, learning_rate
    cfg, sd, tokens, poses, tr = _setup(True, dev)
    for _ in range(3):
        tr.train_step(poses, tokens)
    prefix = ck.save_training_checkpoint(tr, str(tmp_path / 'job'))
    assert os.path.exists(prefix + '.index') and os.path.exists(os.path.join(str(tmp_path / 'job'), 'config.json'))
    ft = finetune_transformer(prefix, total_steps=100, learning_rate=1e-5, device=dev, n_loss_skip=1, pose_multiplier=None)
    assert ft.step_count == 3 and ft.lr_offset == 3 and ft.warmup_steps == 2000 and ft.lr_init == 1e-5 and ft.lr_total_steps == 100
    assert ft.cfg.total_steps == cfg.total_steps and ft.cfg.pose_multiplier == cfg.pose_multiplier
    assert torch.equal(ft.flat_p, tr.flat_p) and torch.equal(ft.flat_m, tr.flat_m) and torch.equal(ft.flat_v, tr.flat_v)
    tr.begin_finetune(learning_rate=1e-5, total_steps=100, warmup_steps=2000)
    tr.train_step(poses, tokens)                       # lr(offset) = 0: moments move, weights do not
    ft.train_step(poses, tokens)
    a, b = tr.train_step(poses, tokens), ft.train_step(poses, tokens)
    assert a['loss'] == b['loss'] and torch.equal(ft.flat_p, tr.flat_p) and torch.equal(ft.flat_m, tr.flat_m)
    assert learning_rate(4, 1e-5, 100, 2000, offset=3) == 1e-5 / 2000
    # an override changes the model that is built (finetune_transformer.py:57-73) ...
    ft2 = finetune_transformer(prefix, total_steps=100, device=dev, pose_multiplier=0.05, weight_decay=0.01)
    assert ft2.cfg.pose_multiplier == 0.05 and ft2.cfg.weight_decay == 0.01 and ft2.step_count == 3
    # ... and a weights-only checkpoint is accepted like .expect_partial() accepts it: optimizer at its initial state
    ck.write_tensor_bundle(str(tmp_path / 'job' / 'weights.only'), ck.state_dict_to_keras(tr.state_dict()))
    ft3 = finetune_transformer(str(tmp_path / 'job' / 'weights.only'), total_steps=100, device=dev)
    assert ft3.step_count == 0 and float(ft3.flat_m.abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        ck.restore_optimizer(ft3, str(tmp_path / 'job' / 'weights.only'), strict=True)


@pytest.mark.gpu
def test_per_tensor_gradient_clipping(dev):
    cfg, sd, tokens, poses, tr = _setup(False, dev, clip=1e-3)
    tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
    for name in tr.names:
        assert float(tr.g(name).norm()) <= 1e-3 * (1 + 1e-4), name


@pytest.mark.gpu
@pytest.mark.parametrize('opts', [dict(random_pose_multiplier=3.0), dict(use_dynamic_pose_loss=True),
                                  dict(random_pose_multiplier=1.7, use_dynamic_pose_loss=True)])
def test_random_pose_multiplier_and_dynamic_pose_loss(dev, opts):
    """migt.py:350-354,160-161 (per-scene position scale in, divided out of the prediction) and :107-120 (learned log-variance
    weighting, summed over the batch): loss and every gradient vs fp64 autograd with the same per-scene factors"""
    from oracle import train_oracle as to
    cfg, sd, tokens, poses, tr = _setup(True, dev, B=3, **opts)
    tr.step_count = 4
    tr.dropout_seed = 11
    if cfg.use_dynamic_pose_loss:
        assert 'pose_loss_weighting_criterion.pos_ori_weights' in tr.names
    metrics = tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
    f = to.random_pose_factors(cfg, 3, tr.step_seed(4)) if cfg.random_pose_multiplier != 1 else None
    if f is not None:
        assert float(f.min()) >= 1 / cfg.random_pose_multiplier - 1e-6 and float(f.max()) <= cfg.random_pose_multiplier + 1e-6
        assert float((f - 1).abs().max()) > 1e-2
    grads, ref = to.gradients(sd, cfg, poses, tokens, step=4, pose_factors=f)
    assert abs(float(metrics['loss']) - ref['loss']) < 1e-4 * max(1.0, abs(ref['loss']))
    assert abs(float(metrics['pose_pos_loss']) - ref['pose_pos_loss']) < 1e-4 * max(1.0, ref['pose_pos_loss'])
    for name in tr.names:
        e = _err(tr.g(name), grads[name].reshape(tr.slices[name][2]))
        assert e < 2e-3, (name, e)
    # the optimizer moves the weighting pair too (it is an ordinary decayed variable, models/utils.py:424)
    if cfg.use_dynamic_pose_loss:
        before = tr.p('pose_loss_weighting_criterion.pos_ori_weights').clone()
        tr.apply_gradients()
        assert float((tr.p('pose_loss_weighting_criterion.pos_ori_weights') - before).abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,S,L,mode', [(2, 2, 3, 64, 'streams'), (1, 3, 4, 64, 'causal'), (2, 1, 5, 16, 'streams'),
                                          (1, 2, 3, 48, 'causal'), (1, 2, 4, 64, 'twin')])
def test_flash_attention_backward_matches_autograd(dev, B, H, S, L, mode):
    """vf_attn_bwd_f32 (probabilities re-materialised from the saved log-sum-exp, masked tiles skipped) against fp64 autograd of
    the oracle's compute_causal_block[_multiend]_attention — every mask mode, tile-aligned (L = 64) and ragged (L = 16, 48) views"""
    from oracle import migt_oracle as mg
    from viewformer_amd import train_ops as T
    d = H * 64
    NS = 3 if mode == 'streams' else 1
    Tn = NS * S * L
    spec = {'causal': -1, 'twin': S - 2, 'streams': -S}[mode]
    g = np.random.Generator(np.random.PCG64(3))
    qkv = torch.from_numpy((g.standard_normal((B * Tn, 3 * d)) * 0.4).astype(np.float32))
    dout = torch.from_numpy(g.standard_normal((B * Tn, d)).astype(np.float32))
    x = qkv.double().requires_grad_(True)
    if mode == 'streams':
        xs = x.view(B, NS, S, L, 3 * d)
        ks, vs, qs = [], [], []
        for s in range(NS):
            v, q, k = xs[:, s].chunk(3, -1)
            ks.append(mg._split_heads(k, H)); vs.append(mg._split_heads(v, H)); qs.append(mg._split_heads(q, H))
        out = torch.stack([mg._merge_heads(a) for a in mg.compute_causal_block_multiend_attention(ks, vs, qs)], 1).reshape(B * Tn, d)
    else:
        v, q, k = x.view(B, S, L, 3 * d).chunk(3, -1)
        sp = lambda t: mg._split_heads(t, H)
        if mode == 'twin':
            # twin views Vc.. : each sees views < Vc and itself -> dense attention with that mask
            view = torch.arange(S).repeat_interleave(L)
            Vc = spec
            vis = (view[None, :] == view[:, None]) | (torch.minimum(view[None, :], torch.tensor(Vc)) < torch.minimum(view[:, None], torch.tensor(Vc)))
            qh, kh, vh = (sp(t).reshape(B, H, S * L, 64) for t in (q, k, v))
            w = qh @ kh.transpose(-1, -2)
            m = vis.double()
            w = w * m - 1e4 * (1 - m)
            a = torch.softmax(w, -1) @ vh
            out = a.permute(0, 2, 1, 3).reshape(B * Tn, d)
        else:
            out = mg._merge_heads(mg.compute_causal_block_attention(sp(k), sp(v), sp(q))).reshape(B * Tn, d)
    out.backward(dout.double())
    ref = x.grad
    gq = qkv.to(dev)
    att = torch.empty((B * Tn, d), device=dev)
    lse = T.attn_fwd_lse(gq[:, d:2 * d], gq[:, 2 * d:], gq[:, :d], att, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, spec)
    assert (att.double().cpu() - out.detach()).abs().max().item() < 3e-5
    dqkv = torch.full((B * Tn, 3 * d), float('nan'), device=dev)
    T.attn_bwd(gq[:, d:2 * d], gq[:, 2 * d:], gq[:, :d], att, dout.to(dev), lse, dqkv[:, d:2 * d], dqkv[:, 2 * d:], dqkv[:, :d],
               B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, spec)
    err = (dqkv.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f'flash attention backward {mode} L={L}: rel err {err:.2e}')
    assert err < 2e-5


@pytest.mark.gpu
def test_flash_and_dense_attention_backward_agree_in_the_train_step(dev):
    grads = {}
    for mode in ('flash', 'dense'):
        cfg, sd, tokens, poses, tr = _setup(True, dev)
        tr.attention_backward = mode
        tr.step_count = 3
        tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
        grads[mode] = tr.flat_g.clone()
    rel = ((grads['flash'] - grads['dense']).abs().max() / grads['dense'].abs().max()).item()
    assert rel < 2e-5, rel


BF16_GRAD_TOL = 6e-2        # (measured worst 3.9e-2, c_attn) per-tensor max |grad error| / max |grad|, dense GEMMs of forward and backward on bf16 MFMA


@pytest.mark.gpu
def test_bf16_training_step_gradients_within_stated_tolerance(dev):
    """the reference trains with --fp16 (mixed_float16, fp32 variables); the bf16 arm of the training step keeps fp32 master
    weights, attention, normalisation, losses and optimizer and runs the dense GEMMs on bf16 MFMA"""
    from oracle import train_oracle as to
    cfg, sd, tokens, poses, tr = _setup(True, dev, precision='bf16')
    assert any(dn.wp16 is not None for dn in tr.model._dense.values()) and len(tr.wpT16) > 0      # the bf16 path really runs
    tr.step_count = 3
    metrics = tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
    grads, ref_metrics = to.gradients(sd, cfg, poses, tokens, step=3)
    assert abs(float(metrics['loss']) - ref_metrics['loss']) < 2e-2 * max(1.0, abs(ref_metrics['loss']))
    worst = ('', 0.0)
    for name in tr.names:
        ref = grads[name].reshape(tr.slices[name][2])
        e = _err(tr.g(name), ref)
        if e > worst[1]:
            worst = (name, e)
        assert e < BF16_GRAD_TOL, (name, e)
    print('bf16 training arm: worst relative gradient error', worst)
    tr.apply_gradients()                                   # the update refreshes the bf16 packings from the fp32 master weights
    m2 = tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
    assert float(m2['loss']) < float(metrics['loss'])


def test_dropout_hash_restatement_statistics():
    """CPU: the numpy restatement of the counter hash behaves like a uniform generator and is index-sensitive"""
    from oracle import train_oracle as to
    idx = np.arange(1 << 18, dtype=np.uint64)
    h = to.dropout_hash(123, 17, idx)
    assert h.dtype == np.uint32
    keep = (h >= np.uint32(int(0.1 * 4294967296.0))).mean()
    assert abs(keep - 0.9) < 3e-3
    assert abs((to.dropout_hash(123, 18, idx) == h).mean()) < 1e-3                 # another site = another mask
    assert abs((to.dropout_hash(124, 17, idx) == h).mean()) < 1e-3                 # another seed = another mask
    hi = to.dropout_hash(123, 17, idx + (np.uint64(1) << np.uint64(32)))           # the high word matters
    assert abs((hi == h).mean()) < 1e-3
    bits = np.unpackbits(h.view(np.uint8)).mean()
    assert abs(bits - 0.5) < 2e-3


def test_dropout_mask_definition_statistics():
    """CPU: the MASK definition (csrc/vf_common.h: four elements share one lowbias32 word, element j keeps iff rotl(word, 8 j) >= thresh):
    the oracle's restatement and the host module's agree; every position of a group has the nominal keep rate; the four decisions of a
    group, neighbouring groups, sites, seeds and mask planes are uncorrelated; the number of dropped elements per group is binomial."""
    from oracle import train_oracle as to
    from viewformer_amd import _hash as hh
    rate, M, N = 0.1, 2048, 768
    thresh = int(rate * 4294967296.0)
    m, n = np.meshgrid(np.arange(M), np.arange(N), indexing='ij')
    g, j = hh.elem_group(m, n, N)
    k = to.dropout_keep(123, 17, g, j, thresh)
    assert np.array_equal(k, hh.dropout_keep(123, 17, g, j, rate))
    sig = np.sqrt(rate * (1 - rate) / k.size)
    assert abs(k.mean() - (1 - thresh / 2 ** 32)) < 4 * sig
    for pos in range(4):                                                           # (a rotation of a uniform word is uniform)
        assert abs(k[pos::4].mean() - 0.9) < 4 * sig * 2
    d = k - k.mean()
    corr = lambda a, b: float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))      # noqa: E731
    lim = 5.0 / np.sqrt(k.size / 4)
    for pos in (1, 2, 3):
        assert abs(corr(d[0::4], d[pos::4])) < lim                                 # inside a group
    assert abs(corr(d[:, :-1], d[:, 1:])) < lim and abs(corr(d[:-4], d[4:])) < lim  # neighbouring groups (next column, next row block)
    for other in (to.dropout_keep(123, 18, g, j, thresh), to.dropout_keep(124, 17, g, j, thresh),
                  to.dropout_keep(123, 17, g + (np.uint64(1) << np.uint64(32)), j, thresh)):   # site, seed, plane (high word)
        assert abs(corr(d, other - other.mean())) < lim
    import math
    nd = (~k.reshape(M // 4, 4, N)).sum(1)
    for c in range(4):
        want = math.comb(4, c) * rate ** c * (1 - rate) ** (4 - c)
        assert abs((nd == c).mean() - want) < 5 * np.sqrt(want / nd.size) + 1e-5, c
    # attention convention: plane (b, h) in the high word, four consecutive keys of a query per group
    ga, ja = hh.attn_group(3, np.arange(130)[:, None], np.arange(130)[None, :], 130)
    assert int(ga[5, 7] >> np.uint64(32)) == 3 and int(ga[5, 7] & np.uint64(0xFFFFFFFF)) == 5 * 33 + 1 and int(ja[5, 7]) == 3
    assert np.array_equal(to.dropout_keep(9, 16, ga, ja, thresh), hh.dropout_keep(9, 16, ga, ja, rate))


@pytest.mark.gpu
@pytest.mark.parametrize('rows,cols', [(1303, 77), (4096, 768), (37, 64)])
def test_dropout_kernels_use_the_restated_mask(dev, rows, cols):
    from oracle import train_oracle as to
    from viewformer_amd import train_ops as T
    rate, seed, site = 0.25, 0xDEADBEEF, 21
    x, r = _rand((rows, cols), 1), _rand((rows, cols), 2)
    m, n = np.meshgrid(np.arange(rows, dtype=np.uint64), np.arange(cols, dtype=np.uint64), indexing='ij')
    keep = to.dropout_keep(seed, site, (m >> np.uint64(2)) * np.uint64(cols) + n, m & np.uint64(3), int(rate * 4294967296.0))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
    want = np.where(keep, x.numpy() * scale, np.float32(0)) + r.numpy()
    got = T.dropout_add(x.to(dev), rate, seed, site, res=r.to(dev))
    assert np.array_equal(got.cpu().numpy(), want.astype(np.float32))
    xd = x.to(dev).clone()
    T.dropout_add(xd, rate, seed, site, out=xd)                                    # in place, no residual
    assert np.array_equal(xd.cpu().numpy(), np.where(keep, x.numpy() * scale, np.float32(0)).astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize('M,K,N', [(1024, 768, 768), (19200, 3072, 768), (1000, 256, 256)])
def test_fused_output_dropout_of_the_projection_gemm(dev, M, K, N):
    """igemm(drop=...): res + dropout(x @ W + b) from ONE epilogue (csrc/gemm_bf16_g256.hip) == the GEMM followed by dropout_add, bit for
    bit (the same accumulator, the same mask, the same fp32 operations in the same order)"""
    from viewformer_amd import ops
    from viewformer_amd import train_ops as T
    g = np.random.Generator(np.random.PCG64(M + N))
    x16 = torch.from_numpy(g.standard_normal((M, K)).astype(np.float32)).to(dev).to(torch.bfloat16)
    w = torch.from_numpy((g.standard_normal((K, N)) * 0.05).astype(np.float32)).to(dev)
    b = torch.from_numpy(g.standard_normal((N,)).astype(np.float32)).to(dev)
    res = torch.from_numpy(g.standard_normal((M, N)).astype(np.float32)).to(dev)
    wp = ops.pack_dense_kn_bf16(w)
    drop = (0.1, 4242, 18)
    assert ops.gemm_drop_supported(M, K, N)
    y = torch.empty((M, N), device=dev)
    ops.igemm(x16, wp, M, K, N, y, bias=b, bf16=True, a16=True)
    want = T.dropout_add(y, *drop, res=res)
    got = torch.full((M, N), float('nan'), device=dev)
    ops.igemm(x16, wp, M, K, N, got, bias=b, res=res, bf16=True, a16=True, drop=drop)
    assert torch.equal(got, want)
    dropped = (got == res).float().mean().item()
    assert abs(dropped - 0.1) < 0.01, dropped
    with pytest.raises(Exception):                                                 # a shape outside the 256-tile kernel is refused, not mis-served
        ops.igemm(x16[:, :64].contiguous(), ops.pack_dense_kn_bf16(w[:64, :128].contiguous()), M, 64, 128, torch.empty((M, 128), device=dev),
                  bf16=True, a16=True, drop=drop)


@pytest.mark.gpu
@pytest.mark.parametrize('rows,d,row0', [(1920, 768, 0), (19200, 768, 0), (515, 256, 0), (1920, 768, 6400), (515, 512, 6), (77, 1024, 12), (64, 768, 3)])
def test_layernorm_backward_masked_bf16_copy(dev, rows, d, row0):
    """layernorm_bwd(also_bf16=True, drop=...): the bf16 copy == bf16(dropout_add(dx)) of that site, while dx, dgamma, dbeta keep their bits.
    (row offsets of a data-parallel shard on and off a mask-group boundary)"""
    from viewformer_amd import train_ops as T
    g = np.random.Generator(np.random.PCG64(rows))
    dy, x, res = (torch.from_numpy(g.standard_normal((rows, d)).astype(np.float32)).to(dev) for _ in range(3))
    gamma = torch.from_numpy((1 + 0.1 * g.standard_normal(d)).astype(np.float32)).to(dev)
    outs = []
    for drop in ((0.0, 0, 0, 0), (0.1, 99, 17, row0)):
        dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        dx, dx16 = T.layernorm_bwd(dy, x, gamma, dg, db, rows, d, res=res, also_bf16=True, drop=drop)
        outs.append((dx, dx16, dg, db))
    (dx0, c0, dg0, db0), (dx1, c1, dg1, db1) = outs
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.equal(c0, dx0.to(torch.bfloat16))
    assert torch.equal(c1, T.dropout_add(dx0, 0.1, 99, 17, row0=row0).to(torch.bfloat16))


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['streams', 'causal'])
def test_flash_attention_with_dropout_matches_autograd(dev, mode):
    """attention-weight dropout inside the forward and both backward kernels == autograd with the same masks"""
    from oracle import migt_oracle as mg
    from oracle import train_oracle as to
    from viewformer_amd import train_ops as T
    B, H, S, L, rate, seed, layer = 2, 2, 3, 64, 0.2, 777, 1
    d = H * 64
    NS = 3 if mode == 'streams' else 1
    Tn = NS * S * L
    spec = -S if mode == 'streams' else -1
    g = np.random.Generator(np.random.PCG64(5))
    qkv = torch.from_numpy((g.standard_normal((B * Tn, 3 * d)) * 0.4).astype(np.float32))
    dout = torch.from_numpy(g.standard_normal((B * Tn, d)).astype(np.float32))
    masks = to.DropoutMasks(rate, seed, B, NS, S, L, d, H)
    x = qkv.double().requires_grad_(True)
    with mg.dropout_masks(masks):
        if mode == 'streams':
            xs = x.view(B, NS, S, L, 3 * d)
            ks, vs, qs = [], [], []
            for s in range(NS):
                v, q, k = xs[:, s].chunk(3, -1)
                ks.append(mg._split_heads(k, H)); vs.append(mg._split_heads(v, H)); qs.append(mg._split_heads(q, H))
            outs = mg.compute_causal_block_multiend_attention(ks, vs, qs, layer=layer)
            out = torch.stack([mg._merge_heads(a) for a in outs], 1).reshape(B * Tn, d)
        else:
            v, q, k = x.view(B, S, L, 3 * d).chunk(3, -1)
            sp = lambda t: mg._split_heads(t, H)
            out = mg._merge_heads(mg.compute_causal_block_attention(sp(k), sp(v), sp(q), wmask=masks.attn(layer, 0, 'main'))).reshape(B * Tn, d)
    out.backward(dout.double())
    gq = qkv.to(dev)
    drop = (rate, seed, 16 + 4 * layer)
    att = torch.empty((B * Tn, d), device=dev)
    lse = T.attn_fwd_lse(gq[:, d:2 * d], gq[:, 2 * d:], gq[:, :d], att, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, spec, drop=drop)
    assert (att.double().cpu() - out.detach()).abs().max().item() < 5e-5
    dqkv = torch.full((B * Tn, 3 * d), float('nan'), device=dev)
    T.attn_bwd(gq[:, d:2 * d], gq[:, 2 * d:], gq[:, :d], att, dout.to(dev), lse, dqkv[:, d:2 * d], dqkv[:, 2 * d:], dqkv[:, :d],
               B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, spec, drop=drop)
    err = (dqkv.double().cpu() - x.grad).abs().max().item() / x.grad.abs().max().item()
    assert err < 2e-5, err


@pytest.mark.gpu
def test_train_step_with_dropout_matches_autograd_with_the_same_masks(dev):
    """the reference's default dropout = 0.1 at all four sites: loss and every gradient vs fp64 autograd over the oracle with the
    build's masks; deterministic per (seed, step); a different step draws different masks"""
    from oracle import train_oracle as to
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    from oracle import migt_oracle as mg
    cfg = MIGTConfig(**TINY_MIGT, dropout=0.1, n_loss_skip=1, localization_weight='1', pose_multiplier=0.2, learning_rate=1e-3,
                     weight_decay=0.05, total_steps=50)
    sd = make_migt_weights(cfg, seed=1, std=0.08)
    g = np.random.Generator(np.random.PCG64(4))
    B, S, t = 2, 4, cfg.token_image_size
    tokens = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(B, S, t, t)))
    _, cams = synthetic_scene_batch(B, S, 8, 1)
    poses = mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(cams))[0])
    tr = MIGTTrainer(MIGT(cfg).load_state_dict(sd).to(dev), warmup_steps=4)
    tr.dropout_seed, tr.step_count = 42, 3
    metrics = tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
    g1 = tr.flat_g.clone()
    grads, ref_metrics = to.gradients_with_dropout(sd, cfg, poses, tokens, 3, 0.1, tr.step_seed(3))
    assert abs(float(metrics['loss']) - ref_metrics['loss']) < 1e-4 * max(1.0, abs(ref_metrics['loss']))
    for name in tr.names:
        e = _err(tr.g(name), grads[name].reshape(tr.slices[name][2]))
        assert e < 2e-3, (name, e)
    tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
    # same (seed, step) -> same masks: identical up to the float-atomic order of the embedding-row scatter (vf_embed_bwd_f32)
    assert ((tr.flat_g - g1).abs().max() / g1.abs().max()).item() < 1e-6
    tr.step_count = 4
    m4 = tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
    assert float(m4['loss']) != float(metrics['loss'])
    # and the no-dropout loss differs from both (dropout really is on)
    _, ref0 = to.gradients(sd, cfg, poses, tokens, step=3)
    assert abs(ref0['loss'] - ref_metrics['loss']) > 1e-4


@pytest.mark.gpu
def test_label_smoothing_matches_autograd(dev):
    from oracle import train_oracle as to
    cfg, sd, tokens, poses, tr = _setup(True, dev, label_smoothing=0.1)
    tr.step_count = 3
    metrics = tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
    grads, ref_metrics = to.gradients(sd, cfg, poses, tokens, step=3)
    _, plain = to.gradients(sd, cfg.__class__(**{**cfg.__dict__, 'label_smoothing': 0.0}) if hasattr(cfg, '__dict__') else cfg, poses, tokens, step=3)
    assert abs(float(metrics['ce_loss']) - ref_metrics['ce_loss']) < 1e-4
    assert abs(ref_metrics['ce_loss'] - plain['ce_loss']) > 1e-3                 # smoothing really changes the loss
    for name in tr.names:
        assert _err(tr.g(name), grads[name].reshape(tr.slices[name][2])) < 2e-3, name


@pytest.mark.gpu
def test_test_step_and_predict_step(dev):
    """MIGT.test_step / predict_step (migt.py:507-541): the compute_losses graph with training=False.  Without dropout and random pose
    multiplier it is the training forward, so its losses equal train_step's; with them on it ignores both (deterministic); the pose
    error metrics follow utils/metrics.py:90-110; predict_step decodes arg-max tokens of every view next to the ground truth."""
    from oracle import migt_oracle as mg
    cfg, sd, tokens, poses, tr = _setup(True, dev)
    tr.step_count = 3
    m_train = tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
    m_test = tr.test_step(poses, tokens)
    for k in ('loss', 'ce_loss', 'acc', 'pose_loss', 'pose_pos_loss', 'pose_ori_loss'):
        assert float(m_test[k]) == float(m_train[k]), k
    # the pose error metrics against the oracle's pose_prediction (LOC stream of the 3-stream graph)
    ref = mg.migt_forward(sd, cfg, tokens, poses, localization_tokens=tokens, output_poses=poses, dtype=torch.float64)
    pp = ref['pose_prediction'][:, cfg.n_loss_skip:].double()
    gt = poses[:, cfg.n_loss_skip:, None].double()
    pos = (pp[..., :3] - gt[..., :3]).norm(dim=-1).mean().item()
    q1 = pp[..., 3:] / pp[..., 3:].norm(dim=-1, keepdim=True)
    q2 = (gt[..., 3:] / gt[..., 3:].norm(dim=-1, keepdim=True)).expand_as(q1)
    w1, v1, w2, v2 = q1[..., :1], q1[..., 1:], q2[..., :1], -q2[..., 1:]
    vec = w1 * v2 + w2 * v1 + torch.cross(v1, v2, dim=-1)
    ori = (2 * torch.asin(vec.norm(dim=-1).clamp(max=1))).mean().item()
    assert abs(float(m_test['pose_pos_err']) - pos) < 1e-4 * max(1.0, pos) and abs(float(m_test['pose_ori_err']) - ori) < 1e-4
    # training-only randomness is off in test_step
    cfg2, _, tokens2, poses2, tr2 = _setup(True, dev)
    tr2.cfg.dropout, tr2.cfg.random_pose_multiplier = 0.3, 2.0
    tr2.step_count = 3                                             # (the localization-weight schedule follows the step counter)
    a, b = tr2.test_step(poses2, tokens2), tr2.test_step(poses2, tokens2)
    assert float(a['loss']) == float(b['loss']) == float(m_test['loss'])
    assert float(tr2.train_step(poses2, tokens2, reduce_gradients=False, apply_update=False)['loss']) != float(a['loss'])
    # PSNR and predict_step: the steps only use the codebook model's duck-typed ``decode_code`` (migt.py:521-526,536-540); a table
    # look-up stands in for it here (the real decoders are covered by the VQGAN tests)
    class Codebook:
        table = (_rand((cfg.n_embeddings, 3), 77) * 0.8).to(dev)

        def decode_code(self, codes):
            assert codes.dtype == torch.int64 and codes.dim() == 3
            return self.table[codes]                                   # [N, t, t, 3] "images" in [-1, 1]
    vq = Codebook()
    m3 = tr.test_step(poses, tokens, codebook_model=vq)
    B, S = tokens.shape[:2]
    gen_last = tr.train_step(poses, tokens, _forward_only=True)[1]['predicted_tokens'][:, -1].reshape(B, 4, 4)
    a, b = (vq.table[gen_last] / 2 + 0.5).clamp(0, 1), (vq.table[tokens[:, -1].to(dev)] / 2 + 0.5).clamp(0, 1)
    want = (10 * torch.log10(1.0 / ((a - b) ** 2).reshape(B, -1).mean(1).clamp_min(1e-12))).mean()
    assert abs(float(m3['psnr']) - float(want)) < 1e-4
    out = tr.predict_step(poses, tokens, vq)
    assert tuple(out['latent_code'].shape) == (B, S, 4, 4) and int(out['latent_code'].max()) < cfg.n_embeddings
    assert tuple(out['decoded_image'].shape) == (B * S, 4, 4, 3) == tuple(out['ground_truth_image'].shape)
    assert torch.equal(out['ground_truth_image'], vq.table[tokens.reshape(-1, 4, 4).to(dev)])
    assert torch.equal(out['decoded_image'], vq.table[out['latent_code'].reshape(-1, 4, 4)])


ATTN_BF16_FWD_TOL, ATTN_BF16_BWD_TOL = 1.5e-2, 3e-2     # max error / max |reference| (bf16 operands incl. P and dS; measured below)


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,S,mode', [(2, 2, 4, 'causal'), (1, 3, 8, 'twin'), (2, 2, 3, 'streams'), (1, 2, 10, 'streams'), (1, 1, 1, 'causal'),
                                        (3, 1, 5, 'twin')])
def test_bf16_flash_attention_forward_lse_and_backward(dev, B, H, S, mode):
    """the bf16 arm's training attention (csrc/attention_dma.hip with its log-sum-exp output, csrc/attention_train_bf16.hip) against
    the exact-f32 kernels on the SAME bf16-rounded q / k / v / dO — which tests/test_train.py pins to fp64 autograd of
    branching_attention.py:5-18,82-126 above.  Every mask mode incl. the training step's 3-stream mask at its real size (S = 10:
    T = 1920, 30 views), workgroups whose second view does not exist, one-view sequences."""
    from viewformer_amd import train_ops as T
    L, d = 64, H * 64
    NS = 3 if mode == 'streams' else 1
    Tn = NS * S * L
    spec = {'causal': -1, 'twin': max(S - 2, 0), 'streams': -S}[mode]
    g = np.random.Generator(np.random.PCG64(17))
    qkv16 = torch.from_numpy((g.standard_normal((B * Tn, 3 * d)) * 0.4).astype(np.float32)).to(dev).to(torch.bfloat16)
    do16 = torch.from_numpy(g.standard_normal((B * Tn, d)).astype(np.float32)).to(dev).to(torch.bfloat16)
    qkv32, do32 = qkv16.float(), do16.float()
    # reference: exact-f32 kernels on the rounded operands
    att32 = torch.empty((B * Tn, d), device=dev)
    lse32 = T.attn_fwd_lse(qkv32[:, d:2 * d], qkv32[:, 2 * d:], qkv32[:, :d], att32, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, spec)
    ref = torch.empty((B * Tn, 3 * d), device=dev)
    T.attn_bwd(qkv32[:, d:2 * d], qkv32[:, 2 * d:], qkv32[:, :d], att32, do32, lse32, ref[:, d:2 * d], ref[:, 2 * d:], ref[:, :d],
               B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, spec)
    # bf16 kernels
    att16 = torch.full((B * Tn, d), float('nan'), dtype=torch.bfloat16, device=dev)
    lse16 = T.attn_fwd_lse_bf16(qkv16[:, d:2 * d], qkv16[:, 2 * d:], qkv16[:, :d], att16, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, spec)
    e_fwd = ((att16.float() - att32).abs().max() / att32.abs().max()).item()
    e_lse = (lse16 - lse32).abs().max().item()
    assert e_fwd < ATTN_BF16_FWD_TOL and e_lse < 5e-3, (e_fwd, e_lse)                  # lse: fp32 softmax over bf16-product scores (q re-rounded
    # to bf16 after the multiplication by scale * log2 e since round 4: scores carry one more 2^-9 rounding)
    got = torch.full((B * Tn, 3 * d), float('nan'), device=dev)
    T.attn_bwd_bf16(qkv16[:, d:2 * d], qkv16[:, 2 * d:], qkv16[:, :d], att16, do16, lse16, got[:, d:2 * d], got[:, 2 * d:], got[:, :d],
                    B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, spec)
    assert not torch.isnan(got).any()
    errs = {}
    for name, sl in (('dv', slice(0, d)), ('dq', slice(d, 2 * d)), ('dk', slice(2 * d, 3 * d))):
        errs[name] = ((got[:, sl] - ref[:, sl]).abs().max() / ref[:, sl].abs().max()).item()
        assert errs[name] < ATTN_BF16_BWD_TOL, (name, errs)
    print(f'bf16 training attention {mode} B={B} H={H} S={S}: fwd {e_fwd:.2e} lse {e_lse:.2e} bwd {errs}')
    # gradients written as bf16 (the c_attn GEMMs' operand format): the fp32 result rounded once
    got16 = torch.full((B * Tn, 3 * d), float('nan'), dtype=torch.bfloat16, device=dev)
    T.attn_bwd_bf16(qkv16[:, d:2 * d], qkv16[:, 2 * d:], qkv16[:, :d], att16, do16, lse16, got16[:, d:2 * d], got16[:, 2 * d:], got16[:, :d],
                    B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, spec)
    assert torch.equal(got16, got.to(torch.bfloat16))
    # deterministic
    got2 = torch.empty_like(got)
    T.attn_bwd_bf16(qkv16[:, d:2 * d], qkv16[:, 2 * d:], qkv16[:, :d], att16, do16, lse16, got2[:, d:2 * d], got2[:, 2 * d:], got2[:, :d],
                    B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, spec)
    assert torch.equal(got, got2)


@pytest.mark.gpu
@pytest.mark.parametrize('BS,L,d,vocab', [(300, 64, 768, 1026), (7, 16, 128, 66), (3, 50, 2048, 10), (4200, 64, 128, 300), (1, 1, 64, 3),
                                           (50, 64, 64, 3000), (20, 16, 64, 12000)])
def test_embedding_backward_with_a_heavy_mask_id(dev, BS, L, d, vocab):
    """vf_embed_bwd_f32 at the training step's shape: a third of the tokens are whole views of the MASK id (the blocks that own it were the
    kernel's whole duration until round 6), ragged token counts, ids outside the table clamped like tf.gather on GPU, a token count past the
    indexed form's range (the scanning kernel), a single token, a vocabulary that leaves the index fewer than 16 chunks and one that leaves it
    none (the scanning kernel again); fixed-order sums — two runs agree bit for bit."""
    from viewformer_amd import train_ops as T
    g = torch.Generator().manual_seed(BS + d)
    ids = torch.randint(0, vocab - 2, (BS, L), generator=g, dtype=torch.int32)
    ids[::3] = vocab - 2                                                # the MASK stream's views
    if BS > 1 and L >= 3:
        ids[1, :3] = torch.tensor([-5, vocab + 7, vocab - 1], dtype=torch.int32)
    dh = torch.randn(BS * L, d, generator=g)
    ref_ids = ids.long().clamp(0, vocab - 1).view(-1)
    wte0, wpe0 = torch.randn(vocab, d, generator=g), torch.randn(L, d, generator=g)
    ref_wte = wte0.double().index_add(0, ref_ids, dh.double())
    ref_wpe = wpe0.double() + dh.double().view(BS, L, d).sum(0)
    ref_add = dh.double().view(BS, L, d).sum(1)
    outs = []
    for _ in range(2):
        dwte, dwpe = wte0.clone().to(dev), wpe0.clone().to(dev)
        dadd = T.embed_bwd(dh.to(dev), ids.view(-1).to(dev), dwte, dwpe, BS, L, d, vocab)
        outs.append((dwte, dwpe, dadd))
    tol = 2e-6 * max(1.0, (BS * L / 19200.0) ** 0.5)                    # fp32 running sums: the bound grows with the root of the term count
    errs = (_err(outs[0][0], ref_wte), _err(outs[0][1], ref_wpe), _err(outs[0][2].view(BS, d), ref_add))
    assert max(errs) < tol, (errs, tol)
    for a, b in zip(*outs):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize('rows,K,N', [(6400, 1536, 7), (130, 1536, 7), (64, 256, 8), (1000, 768, 3), (1, 4, 1)])
def test_small_n_dense_layer_and_its_weight_gradient(dev, rows, K, N):
    """round 6: the pose head's 1536 -> 7 layer (migt.py:291-292,354) and its dW on one-pass kernels (csrc/train_ops.hip: dense_small_n*) against
    fp64; deterministic (slab partials folded in slab order); accumulate adds; the trainer's two paths (small_n_pose_head on / off) give the same
    step within fp32 rounding (test_small_n_pose_head_matches_the_implicit_gemm_path below)."""
    from viewformer_amd import train_ops as T
    g = np.random.Generator(np.random.PCG64(61))
    x = torch.from_numpy(g.standard_normal((rows, K)).astype(np.float32)).to(dev)
    W = torch.from_numpy((g.standard_normal((K, N)) * 0.05).astype(np.float32)).to(dev)
    b = torch.from_numpy(g.standard_normal(N).astype(np.float32)).to(dev)
    dy = torch.from_numpy(g.standard_normal((rows, N)).astype(np.float32)).to(dev)
    assert T.dense_small_n_supported(K, N)
    out = T.dense_small_n(x, W, b, rows, K, N)
    ref = x.double() @ W.double() + b.double()
    mag = x.double().abs() @ W.double().abs() + b.double().abs()
    assert float(((out.double() - ref).abs() / mag).max()) < 2e-6
    assert torch.equal(out, T.dense_small_n(x, W, b, rows, K, N))
    assert torch.equal(T.dense_small_n(x, W, None, rows, K, N), T.dense_small_n(x, W, torch.zeros_like(b), rows, K, N))
    dW = torch.full((K, N), float('nan'), device=dev)
    T.dense_small_n_wgrad(x, dy, dW, rows, K, N, accumulate=False)
    refw = x.double().T @ dy.double()
    magw = x.double().abs().T @ dy.double().abs()
    assert float(((dW.double() - refw).abs() / magw.clamp_min(1e-30)).max()) < 4e-6
    dW2 = dW.clone()
    T.dense_small_n_wgrad(x, dy, dW2, rows, K, N, accumulate=True)
    assert torch.allclose(dW2, 2 * dW, rtol=1e-5, atol=1e-5 * float(dW.abs().max()))      # (dst + fold(slabs): one more rounding per element)
    dW3 = torch.empty_like(dW)
    T.dense_small_n_wgrad(x, dy, dW3, rows, K, N, accumulate=False)
    assert torch.equal(dW3, dW)                                     # deterministic
    # a strided x (a column slice of a wider activation): ldx > K
    if K >= 256:
        wide = torch.cat([x, x.flip(1)], 1)
        assert torch.equal(T.dense_small_n(wide[:, :K], W, b, rows, K, N), out)


@pytest.mark.gpu
def test_small_n_pose_head_matches_the_implicit_gemm_path(dev):
    """the training step with the pose head's 1536 -> 7 layer on the small-N kernels (default) and on the implicit-GEMM kernel (round 5): same loss,
    same gradients within fp32 summation-order noise — incl. the pose head's own weight and bias gradients"""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    from oracle import migt_oracle as mg
    cfg = MIGTConfig(n_layer=2, d_model=256, n_head=4, sequence_size=4, n_loss_skip=1, localization_weight='2', pose_multiplier=0.2, dropout=0.0,
                     learning_rate=1e-3, weight_decay=0.05, total_steps=50)
    sd = make_migt_weights(cfg, seed=4)
    g = np.random.Generator(np.random.PCG64(12))
    B, S = 2, 4
    tokens = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(B, S, 8, 8)))
    _, cams = synthetic_scene_batch(B, S, 8, 6)
    poses = mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(cams))[0])
    res = {}
    for small in (True, False):
        tr = MIGTTrainer(MIGT(cfg).load_state_dict(sd).to(dev), warmup_steps=4)
        tr.small_n_pose_head = small
        m = tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
        res[small] = (float(m['loss']), tr.flat_g.clone(), tr)
    assert abs(res[True][0] - res[False][0]) < 1e-5 * max(1.0, abs(res[False][0]))
    tr = res[True][2]
    for n in tr.names:
        a, b, _ = tr.slices[n]
        ref = res[False][1][a:b]
        if float(ref.abs().max()) > 0:
            e = ((res[True][1][a:b] - ref).abs().max() / ref.abs().max()).item()
            assert e < 2e-5, (n, e)


@pytest.mark.gpu
def test_one_launch_repack_equals_the_per_tensor_packings(dev):
    """vf_gemm_bf16_pack_multi (the training step's one-launch refresh of every layer's W and W^T packing; round 6: one thread per packed 16-byte
    group, coalesced loads — 211 instead of 321 us for the step's table) against vf_gemm_bf16_pack, bit for bit: the transformer's four layer
    shapes, both orientations, and shapes that end inside a 64-deep chunk / a 128-wide block (zero padding)."""
    from viewformer_amd import ops
    g = np.random.Generator(np.random.PCG64(5))
    items, refs = [], []
    for r, c in ((768, 2304), (768, 768), (768, 3072), (3072, 768), (1026, 768), (200, 136), (64, 128)):
        w = torch.from_numpy(g.standard_normal((r, c)).astype(np.float32)).to(dev)
        for tr in (False, True):
            ref = ops.pack_dense_nk_bf16(w) if tr else ops.pack_dense_kn_bf16(w)
            items.append((w, tr, torch.full_like(ref, float('nan'))))
            refs.append(ref)
    run = ops.pack_bf16_multi(items)
    run()                                                        # (the closure re-runs the same launch: the step's refresh)
    torch.cuda.synchronize()
    for (w, tr, out), ref in zip(items, refs):
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16)), (tuple(w.shape), tr)


ATTN_BF16_BWD_TOL_PEAKED = 6e-2      # the same bound at score magnitudes of a trained model (measured: see profiles/r6_attention_diet.txt)


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,S,mode', [(2, 2, 3, 'streams'), (2, 2, 4, 'causal')])
def test_bf16_flash_attention_backward_at_trained_scale_scores(dev, B, H, S, mode):
    """MIGT's scores are UNSCALED (branching_attention.py:5-18), so a bf16 rounding of q (2^-9 relative) moves a score by |s| 2^-9 —
    every other attention test of this file runs at init-scale weights (|s| < 5).  Until round 6 the bf16 forward rounded q' = bf16(q log2 e)
    while the backward re-materialised P from the un-rounded q; now the dQ kernel uses the forward's q' (its P is the forward's), the dK / dV
    kernel still streams the un-rounded q.  Here |s| reaches 25-40 (a peaked, trained-model softmax): the gradients must stay within the stated bound of the
    exact-f32 kernels on the same operands, the forward within its own."""
    from viewformer_amd import train_ops as T
    L, d = 64, H * 64
    NS = 3 if mode == 'streams' else 1
    Tn = NS * S * L
    spec = {'causal': -1, 'streams': -S}[mode]
    g = np.random.Generator(np.random.PCG64(41))
    qkv16 = torch.from_numpy(g.standard_normal((B * Tn, 3 * d)).astype(np.float32)).to(dev).to(torch.bfloat16)       # s = q.k: std 8
    do16 = torch.from_numpy(g.standard_normal((B * Tn, d)).astype(np.float32)).to(dev).to(torch.bfloat16)
    qkv32, do32 = qkv16.float(), do16.float()
    smax = float((qkv32[:Tn, d:d + 64] @ qkv32[:Tn, 2 * d:2 * d + 64].T).abs().max())
    assert smax > 25.0, smax
    att32 = torch.empty((B * Tn, d), device=dev)
    lse32 = T.attn_fwd_lse(qkv32[:, d:2 * d], qkv32[:, 2 * d:], qkv32[:, :d], att32, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, spec)
    ref = torch.empty((B * Tn, 3 * d), device=dev)
    T.attn_bwd(qkv32[:, d:2 * d], qkv32[:, 2 * d:], qkv32[:, :d], att32, do32, lse32, ref[:, d:2 * d], ref[:, 2 * d:], ref[:, :d],
               B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, spec)
    att16 = torch.full((B * Tn, d), float('nan'), dtype=torch.bfloat16, device=dev)
    lse16 = T.attn_fwd_lse_bf16(qkv16[:, d:2 * d], qkv16[:, 2 * d:], qkv16[:, :d], att16, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, spec)
    e_fwd = ((att16.float() - att32).abs().max() / att32.abs().max()).item()
    got = torch.full((B * Tn, 3 * d), float('nan'), device=dev)
    T.attn_bwd_bf16(qkv16[:, d:2 * d], qkv16[:, 2 * d:], qkv16[:, :d], att16, do16, lse16, got[:, d:2 * d], got[:, 2 * d:], got[:, :d],
                    B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, spec)
    assert not torch.isnan(got).any()
    errs = {name: ((got[:, sl] - ref[:, sl]).abs().max() / ref[:, sl].abs().max()).item()
            for name, sl in (('dv', slice(0, d)), ('dq', slice(d, 2 * d)), ('dk', slice(2 * d, 3 * d)))}
    print(f'bf16 training attention at |s| <= {smax:.0f} ({mode}): fwd {e_fwd:.2e} bwd {errs}')
    from conftest import parity_report
    parity_report(test='bf16_attention_trained_scale', mode=mode, max_abs_score=smax, fwd_err=e_fwd, **errs)
    assert e_fwd < 4 * ATTN_BF16_FWD_TOL, e_fwd
    assert max(errs.values()) < ATTN_BF16_BWD_TOL_PEAKED, errs


@pytest.mark.gpu
@pytest.mark.parametrize('B,H,S,mode', [(2, 2, 4, 'causal'), (2, 2, 3, 'streams'), (1, 2, 10, 'streams'), (3, 1, 5, 'twin')])
def test_bf16_flash_attention_with_dropout(dev, B, H, S, mode):
    """attention dropout inside the bf16 forward / dQ / dK-dV kernels (round 4): the SAME masks as the exact-f32 kernels — which
    test_flash_attention_with_dropout_matches_autograd pins to fp64 autograd with the masks restated in numpy — so on the same
    bf16-rounded operands the two arms agree within the bf16 arm's own tolerance; the log-sum-exp is over the undropped weights, i.e.
    bit-identical to the no-dropout call; the fraction of exactly-zero probabilities is the rate (checked through P.V with V = I)."""
    from viewformer_amd import train_ops as T
    L, d = 64, H * 64
    NS = 3 if mode == 'streams' else 1
    Tn = NS * S * L
    spec = {'causal': -1, 'twin': max(S - 2, 0), 'streams': -S}[mode]
    drop = (0.1, 31337, 24)
    g = np.random.Generator(np.random.PCG64(23))
    qkv16 = torch.from_numpy((g.standard_normal((B * Tn, 3 * d)) * 0.4).astype(np.float32)).to(dev).to(torch.bfloat16)
    do16 = torch.from_numpy(g.standard_normal((B * Tn, d)).astype(np.float32)).to(dev).to(torch.bfloat16)
    qkv32, do32 = qkv16.float(), do16.float()
    att32 = torch.empty((B * Tn, d), device=dev)
    lse32 = T.attn_fwd_lse(qkv32[:, d:2 * d], qkv32[:, 2 * d:], qkv32[:, :d], att32, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, spec, drop=drop)
    ref = torch.empty((B * Tn, 3 * d), device=dev)
    T.attn_bwd(qkv32[:, d:2 * d], qkv32[:, 2 * d:], qkv32[:, :d], att32, do32, lse32, ref[:, d:2 * d], ref[:, 2 * d:], ref[:, :d],
               B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, spec, drop=drop)
    att16 = torch.full((B * Tn, d), float('nan'), dtype=torch.bfloat16, device=dev)
    lse16 = T.attn_fwd_lse_bf16(qkv16[:, d:2 * d], qkv16[:, 2 * d:], qkv16[:, :d], att16, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, spec, drop=drop)
    att_nd = torch.empty_like(att16)
    lse_nd = T.attn_fwd_lse_bf16(qkv16[:, d:2 * d], qkv16[:, 2 * d:], qkv16[:, :d], att_nd, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, spec)
    assert torch.equal(lse16, lse_nd) and not torch.equal(att16, att_nd)
    e_fwd = ((att16.float() - att32).abs().max() / att32.abs().max()).item()
    assert e_fwd < ATTN_BF16_FWD_TOL, e_fwd
    got = torch.full((B * Tn, 3 * d), float('nan'), device=dev)
    T.attn_bwd_bf16(qkv16[:, d:2 * d], qkv16[:, 2 * d:], qkv16[:, :d], att16, do16, lse16, got[:, d:2 * d], got[:, 2 * d:], got[:, :d],
                    B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, spec, drop=drop)
    assert not torch.isnan(got).any()
    errs = {}
    for name, sl in (('dv', slice(0, d)), ('dq', slice(d, 2 * d)), ('dk', slice(2 * d, 3 * d))):
        errs[name] = ((got[:, sl] - ref[:, sl]).abs().max() / ref[:, sl].abs().max()).item()
        assert errs[name] < ATTN_BF16_BWD_TOL, (name, errs)
    print(f'bf16 training attention with dropout {mode} B={B} H={H} S={S}: fwd {e_fwd:.2e} bwd {errs}')
    got2 = torch.empty_like(got)
    T.attn_bwd_bf16(qkv16[:, d:2 * d], qkv16[:, 2 * d:], qkv16[:, :d], att16, do16, lse16, got2[:, d:2 * d], got2[:, 2 * d:], got2[:, :d],
                    B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, spec, drop=drop)
    assert torch.equal(got, got2)                                                  # deterministic
    # the mask itself: q = k = 0 makes every visible weight equal, V = one-hot(key position within its view) -> the output of query i is
    # (1 / (1 - rate)) / (visible keys) * (kept keys at each position): exactly-zero entries of a single-view causal call mark dropped keys
    if mode == 'causal' and S == 4:
        from oracle import train_oracle as to
        z = torch.zeros((B * Tn, 3 * d), dtype=torch.bfloat16, device=dev)
        eye = torch.eye(64, dtype=torch.bfloat16, device=dev).repeat(B * S, H)      # V[t][h*64 + c] = (t % 64 == c)
        z[:, :d] = eye
        o = torch.empty((B * Tn, d), dtype=torch.bfloat16, device=dev)
        T.attn_fwd_lse_bf16(z[:, d:2 * d], z[:, 2 * d:], z[:, :d], o, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, spec, drop=drop)
        o = o.float().view(B, Tn, H, 64).cpu().numpy()
        thresh = int(drop[0] * 4294967296.0)
        for b in range(B):
            for h in range(H):
                q = np.arange(64)                                                  # queries of view 0 see exactly the 64 keys of view 0
                qq, kk = np.meshgrid(q, np.arange(64), indexing='ij')
                grp = (np.uint64(b * H + h) << np.uint64(32)) | (qq.astype(np.uint64) * np.uint64(Tn // 4) + (kk.astype(np.uint64) >> np.uint64(2)))
                keep = to.dropout_keep(drop[1], drop[2], grp, kk.astype(np.uint64) & np.uint64(3), thresh)
                assert np.array_equal(o[b, :64, h] != 0, keep), (b, h)


@pytest.mark.gpu
@pytest.mark.parametrize('dropout', [0.0, 0.1])
def test_bf16_training_arm_at_widths_the_tn_kernel_does_not_tile(dev, dropout):
    """d_model = 384 (head dim 64, a multiple of 128 but not of 256): the bf16 packings and the bf16 attention apply, the TN weight-gradient
    kernel and the 256-tile GEMM do not (K, N % 256).  ADVICE r3: the step then handed a bf16 gradient to the transpose + pack path and
    raised.  Since round 4 the bf16 gradient operands are switched on by an explicit predicate over the four layer shapes: here they stay
    fp32 and the step runs — gradients within the bf16 arm's tolerance of the fp32-equivalent arm, with and without dropout (fused where
    the 256-tile kernel applies, the separate passes where it does not: same masks)."""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    from viewformer_amd import ops
    from oracle import migt_oracle as mg
    cfg = MIGTConfig(n_layer=2, d_model=384, n_head=6, sequence_size=4, n_loss_skip=1, localization_weight='2', pose_multiplier=0.2,
                     dropout=dropout, learning_rate=1e-3, weight_decay=0.05, total_steps=50)
    sd = make_migt_weights(cfg, seed=4)                            # (the reference's initializer range, 0.02: with std 0.05 this toy's logits
    # saturate — loss 10.9 against ln 1024 = 6.9 — and the bf16 arm sits at 4.9e-2 .. 7.7e-2 of the fp32 arm, dropout or not: tools/diag384.py)
    g = np.random.Generator(np.random.PCG64(11))
    B, S = 2, 4                                                   # M = 2 * 3 * 4 * 64 = 1536 rows
    tokens = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(B, S, 8, 8)))
    _, cams = synthetic_scene_batch(B, S, 8, 5)
    poses = mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(cams))[0])
    assert not ops.gemm_tn_bf16_shape_ok(1536, 384, 1152)
    grads, losses = {}, {}
    for arm in ('bf16', 'f32'):
        tr = MIGTTrainer(MIGT(cfg, precision=arm).load_state_dict(sd).to(dev), warmup_steps=4)
        tr.step_count, tr.dropout_seed = 3, 17
        m = tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
        grads[arm], losses[arm] = tr.flat_g.clone(), float(m['loss'])
        assert bool(torch.isfinite(tr.flat_g).all())
    assert abs(losses['bf16'] - losses['f32']) < 2e-2 * max(1.0, abs(losses['f32']))
    worst = 0.0
    for n in tr.names:
        a, b, _ = tr.slices[n]
        ref = grads['f32'][a:b]
        if float(ref.abs().max()) > 0:
            worst = max(worst, ((grads['bf16'][a:b] - ref).abs().max() / ref.abs().max()).item())
    print(f'd_model 384, dropout {dropout}: bf16 arm vs fp32-equivalent arm, worst per-tensor gradient difference', worst)
    assert worst < BF16_GRAD_TOL, worst


@pytest.mark.gpu
def test_bf16_arm_lm_head_backward_uses_current_weights_when_rows_are_not_a_multiple_of_64(dev):
    """with d_model and n_embeddings multiples of 256 the bf16 arm keeps bf16 LM-head packings, but the step only uses
    them when M1 = B*S*L is a multiple of 64; otherwise dH = dlogits @ wte runs on the native transposed packing — which repack() no longer
    refreshed, so from step 2 on it held step-1 weights with no error raised.  Here M1 = 2*3*16 = 96 (a multiple of the native GEMM's 32-row
    K stage, not of 64): after real optimizer steps the
    third step's gradients must (a) equal, bit for bit, those of a FRESH trainer built from the updated weights and (b) sit within the bf16
    arm's tolerance of fp64 autograd over the oracle at those weights."""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    from oracle import migt_oracle as mg
    from oracle import train_oracle as to
    cfg = MIGTConfig(n_embeddings=256, n_head=4, d_model=256, n_layer=2, token_image_size=4, sequence_size=3, dropout=0.0, n_loss_skip=1,
                     localization_weight='2', pose_multiplier=0.2, learning_rate=2e-2, weight_decay=0.05, total_steps=50)
    sd = make_migt_weights(cfg, seed=6, std=0.08)
    g = np.random.Generator(np.random.PCG64(21))
    B, S = 2, 3
    tokens = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(B, S, 4, 4)))
    _, cams = synthetic_scene_batch(B, S, 8, 7)
    poses = mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(cams))[0])
    tr = MIGTTrainer(MIGT(cfg, precision='bf16').load_state_dict(sd).to(dev), warmup_steps=1)
    assert tr._lm16 is not None and (B * S * 16) % 64 != 0         # bf16 LM-head packings exist, and this batch cannot use them
    for _ in range(3):
        tr.train_step(poses, tokens)                               # three updates (the first at warm-up lr 0; lr 2e-2: the weights really move)
    moved = (tr.state_dict()['wte.weight'].cpu() - torch.as_tensor(sd['wte.weight'])).abs().max().item()
    assert moved > 1e-2, moved
    tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
    got = tr.flat_g.clone()
    sd2 = {k: v.cpu().numpy() for k, v in tr.state_dict().items()}
    fresh = MIGTTrainer(MIGT(cfg, precision='bf16').load_state_dict(sd2).to(dev), warmup_steps=1)
    fresh.step_count = tr.step_count
    fresh.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
    assert torch.equal(got, fresh.flat_g)                          # a stale transposed LM head fails here
    grads, _ = to.gradients(sd2, cfg, poses, tokens, step=tr.step_count)
    for name in tr.names:
        ref = grads[name].reshape(tr.slices[name][2])
        assert _err(tr.g(name), ref) < BF16_GRAD_TOL, name


@pytest.mark.gpu
def test_bf16_attention_is_what_the_bf16_training_arm_runs_at_full_width(dev):
    """head dim 64, 64-token views, no dropout: the trainer takes the bf16 attention kernels; with attention_arith='f32' the exact-f32
    ones — gradients of the two agree within the bf16 arm's own tolerance"""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    from oracle import migt_oracle as mg
    cfg = MIGTConfig(n_layer=2, d_model=256, n_head=4, sequence_size=4, n_loss_skip=1, localization_weight='2', pose_multiplier=0.2,
                     dropout=0.0, learning_rate=1e-3, weight_decay=0.05, total_steps=50)
    sd = make_migt_weights(cfg, seed=4, std=0.05)
    g = np.random.Generator(np.random.PCG64(9))
    B, S = 2, 4                                                   # M = 2 * 3 * 4 * 64 = 1536 rows
    tokens = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(B, S, 8, 8)))
    _, cams = synthetic_scene_batch(B, S, 8, 5)
    poses = mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(cams))[0])
    grads = {}
    for arith in ('bf16', 'f32'):
        tr = MIGTTrainer(MIGT(cfg, precision='bf16').load_state_dict(sd).to(dev), warmup_steps=4)
        tr.attention_arith = arith
        tr.step_count = 3
        calls = []
        from viewformer_amd import train_ops as T
        orig = T.attn_bwd_bf16
        T.attn_bwd_bf16 = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            tr.train_step(poses, tokens, reduce_gradients=False, apply_update=False)
        finally:
            T.attn_bwd_bf16 = orig
        assert (len(calls) == cfg.n_layer) == (arith == 'bf16')
        grads[arith] = tr.flat_g.clone()
    worst = 0.0
    for n in tr.names:
        a, b, _ = tr.slices[n]
        ref = grads['f32'][a:b]
        if float(ref.abs().max()) > 0:
            worst = max(worst, ((grads['bf16'][a:b] - ref).abs().max() / ref.abs().max()).item())
    print('bf16 vs f32 attention inside the bf16 training arm: worst per-tensor gradient difference', worst)
    assert worst < BF16_GRAD_TOL, worst


@pytest.mark.gpu
@pytest.mark.parametrize('M,K,N', [(64, 256, 256), (640, 512, 256), (1920, 256, 768), (19200, 768, 2304)])
def test_tn_weight_gradient_gemm(dev, M, K, N):
    """csrc/gemm_tn_bf16.hip: dW = x^T dy and db = column sums of dy from the row-major bf16 activation and fp32 gradient, against fp64 on
    the operands it multiplies (x as stored, dy rounded to bf16 — products of bf16 values are exact in fp32, so only the accumulation
    order differs) and against the path it replaces (widening transpose + bf16 packing + batched split-K GEMM + column sums)."""
    from viewformer_amd import ops
    from viewformer_amd import train_ops as T
    g = np.random.Generator(np.random.PCG64(M + K))
    x16 = torch.from_numpy(g.standard_normal((M, K)).astype(np.float32)).to(dev).to(torch.bfloat16)
    dy = torch.from_numpy((g.standard_normal((M, N)) * 0.1).astype(np.float32)).to(dev)
    assert ops.gemm_tn_bf16_supported(x16, M, K, N)
    dw0 = torch.from_numpy(g.standard_normal((K, N)).astype(np.float32)).to(dev)
    db0 = torch.from_numpy(g.standard_normal((N,)).astype(np.float32)).to(dev)
    dw, db = dw0.clone(), db0.clone()
    ops.gemm_tn_bf16(x16, dy, M, K, N, dw, db)                                  # accumulates
    ref_w = x16.double().T @ dy.to(torch.bfloat16).double()
    ref_b = dy.double().sum(0)
    e_w = ((dw.double() - dw0.double() - ref_w).abs().max() / ref_w.abs().max()).item()
    e_b = ((db.double() - db0.double() - ref_b).abs().max() / ref_b.abs().max()).item()
    assert e_w < 2e-5 and e_b < 2e-5, (e_w, e_b)
    dw2, db2 = dw0.clone(), db0.clone()
    ops.gemm_tn_bf16(x16, dy, M, K, N, dw2, db2)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)                        # deterministic
    flat = torch.cat([dw0.reshape(-1), db0])                                    # the trainer's layout (bias right behind the weight): ONE slab sum
    ops.gemm_tn_bf16(x16, dy, M, K, N, flat[:K * N].view(K, N), flat[K * N:])
    assert torch.equal(flat[:K * N].view(K, N), dw) and torch.equal(flat[K * N:], db)
    # the gradient arriving already rounded to bf16 (its producer wrote the rounding this kernel applies): the same products in the
    # same order -> the same dW bit for bit; db becomes the sum of the rounded values
    dw3, db3 = dw0.clone(), db0.clone()
    ops.gemm_tn_bf16(x16, dy.to(torch.bfloat16), M, K, N, dw3, db3)
    assert torch.equal(dw3, dw)
    ref_b16 = dy.to(torch.bfloat16).double().sum(0)
    assert ((db3.double() - db0.double() - ref_b16).abs().max() / ref_b16.abs().max()).item() < 2e-5
    if M % 128 == 0:                                                            # the replaced path, same operands
        xt = T.transpose(x16, M, K)
        old = torch.zeros((K, N), device=dev)
        ops.igemm(xt, ops.pack_dense_kn_bf16(dy), K, M, N, old, lda=M, bf16=True)
        assert ((old.double() - ref_w).abs().max() / ref_w.abs().max()).item() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('M', [512, 640])         # whole 256-row tiles / a ragged last tile
def test_gelu_backward_epilogue_of_the_256_tile_kernel(dev, M):
    """VF_EPI_GELU_BWD on the 256 x 256 LDS-DMA kernel (bf16 gradient in, bf16 d(pre-activation) out): the same bits as the 128-tile
    kernel's epilogue (vf_select(VF_SEL_GEMM_G256, 0)) and as the separate passes (fp32 dX, then gelu_bwd with a bf16 result)."""
    from viewformer_amd import ops
    from viewformer_amd import train_ops as T
    K, N = 768, 3072                                                             # dy [M, K] @ W^T [K, N]
    g = np.random.Generator(np.random.PCG64(M))
    dy16 = torch.from_numpy((g.standard_normal((M, K)) * 0.1).astype(np.float32)).to(dev).to(torch.bfloat16)
    u = torch.from_numpy((g.standard_normal((M, N)) * 1.5).astype(np.float32)).to(dev)
    w = torch.from_numpy((g.standard_normal((K, N)) * 0.05).astype(np.float32)).to(dev)
    wp = ops.pack_dense_kn_bf16(w)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.igemm(dy16, wp, M, K, N, out, res=u, epilogue=ops.EPI_GELU_BWD, bf16=True, a16=True, o16=True)
    from viewformer_amd import _lib
    prev = _lib.select(_lib.SEL_GEMM_G256, 0)
    try:
        out128 = torch.empty_like(out)
        ops.igemm(dy16, wp, M, K, N, out128, res=u, epilogue=ops.EPI_GELU_BWD, bf16=True, a16=True, o16=True)
        dx = torch.empty((M, N), device=dev)
        ops.igemm(dy16, wp, M, K, N, dx, bf16=True, a16=True)
    finally:
        _lib.select(_lib.SEL_GEMM_G256, prev)
    assert torch.equal(out, out128)
    assert torch.equal(out, T.gelu_bwd(u, dx, out_bf16=True))
    ref = dy16.double() @ w.to(torch.bfloat16).double()
    x = u.double()
    gp = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * np.pi) ** 0.5
    err = ((out.double() - ref * gp).abs().max() / (ref * gp).abs().max()).item()
    assert err < 6e-3, err                                                       # bf16 output rounding (2^-9) + the fast erf


@pytest.mark.gpu
def test_layernorm_backward_bf16_copy(dev):
    from viewformer_amd import train_ops as T
    M, d = 1000, 768
    g = np.random.Generator(np.random.PCG64(5))
    dy, x, res = (torch.from_numpy(g.standard_normal((M, d)).astype(np.float32)).to(dev) for _ in range(3))
    gamma = torch.from_numpy(g.standard_normal(d).astype(np.float32)).to(dev)
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    dx = T.layernorm_bwd(dy, x, gamma, dg, db, M, d, res=res)
    dg2, db2 = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    dx2, dx16 = T.layernorm_bwd(dy, x, gamma, dg2, db2, M, d, res=res, also_bf16=True)
    assert torch.equal(dx, dx2) and torch.equal(dg, dg2) and torch.equal(db, db2)
    assert torch.equal(dx16, dx.to(torch.bfloat16))


@pytest.mark.gpu
@pytest.mark.parametrize('M', [512, 640])
def test_gelu_dual_epilogue_of_the_256_tile_kernel(dev, M):
    """VF_EPI_GELU_DUAL: the fp32 pre-activation is the plain GEMM's (same bits) and the bf16 GELU beside it is the inference arm's fused
    c_fc epilogue (bias + fast-erf GELU, bf16 store: same bits); against the library-erf pass a small fraction of outputs differ, by one bf16 step."""
    from viewformer_amd import ops, _lib
    from viewformer_amd import train_ops as T
    K, N = 768, 3072
    g = np.random.Generator(np.random.PCG64(M + 7))
    x16 = torch.from_numpy(g.standard_normal((M, K)).astype(np.float32)).to(dev).to(torch.bfloat16)
    w = torch.from_numpy((g.standard_normal((K, N)) * 0.05).astype(np.float32)).to(dev)
    b = torch.from_numpy(g.standard_normal(N).astype(np.float32)).to(dev)
    wp = ops.pack_dense_kn_bf16(w)
    u = torch.empty((M, N), device=dev)
    f = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.igemm(x16, wp, M, K, N, u, bias=b, epilogue=ops.EPI_GELU_DUAL, bf16=True, a16=True, out_aux=f)
    u0 = torch.empty_like(u)
    ops.igemm(x16, wp, M, K, N, u0, bias=b, bf16=True, a16=True)
    f0 = torch.empty_like(f)
    ops.igemm(x16, wp, M, K, N, f0, bias=b, epilogue=ops.EPI_GELU, bf16=True, a16=True, o16=True)
    assert torch.equal(u, u0) and torch.equal(f, f0)
    fp = T.gelu(u, out_bf16=True)                                               # library erff
    # (the fast erf is 1.5e-7 ABSOLUTE from erff: in GELU's negative tail, where the result is ~1e-3 and below, that moves a bf16 rounding
    # often — 0.2 % of all outputs differ, each by one bf16 step)
    ne = (f != fp)
    assert float(ne.float().mean()) < 5e-3, float(ne.float().mean())
    assert bool(((f.float() - fp.float()).abs() <= fp.float().abs() * 2 ** -7 + 2e-7).all())
    # the pre-activation itself as bf16 (the training arm's default): the rounded fp32 value; the GELU beside it is unchanged (it is taken
    # from the fp32 accumulator, not from the rounded u)
    u16 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    f2 = torch.empty_like(f)
    ops.igemm(x16, wp, M, K, N, u16, bias=b, epilogue=ops.EPI_GELU_DUAL, bf16=True, a16=True, o16=True, out_aux=f2)
    assert torch.equal(u16, u.to(torch.bfloat16)) and torch.equal(f2, f)
    # ... and the GELU-backward epilogue reading that bf16 u: the bits of the same epilogue on the widened copy
    dy16 = torch.from_numpy((g.standard_normal((M, K)) * 0.1).astype(np.float32)).to(dev).to(torch.bfloat16)
    wt = ops.pack_dense_kn_bf16(torch.from_numpy((g.standard_normal((K, N)) * 0.05).astype(np.float32)).to(dev))
    du_a = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    du_b = torch.empty_like(du_a)
    ops.igemm(dy16, wt, M, K, N, du_a, res=u16, epilogue=ops.EPI_GELU_BWD, bf16=True, a16=True, o16=True, res16=True)
    ops.igemm(dy16, wt, M, K, N, du_b, res=u16.float(), epilogue=ops.EPI_GELU_BWD, bf16=True, a16=True, o16=True)
    assert torch.equal(du_a, du_b)
    with pytest.raises(_lib.VfError):                                           # the aux output needs its epilogue, and the reverse
        ops.igemm(x16, wp, M, K, N, u, bias=b, bf16=True, a16=True, out_aux=f)
    with pytest.raises(_lib.VfError):
        ops.igemm(x16, wp, M, K, N, u, bias=b, epilogue=ops.EPI_GELU_DUAL, bf16=True, a16=True)


@pytest.mark.gpu
@pytest.mark.parametrize('drop', [(0.0, 0, 0), (0.1, 9, 3)])
def test_attention_backward_launches_on_two_streams_give_the_one_stream_results(dev, drop):
    """vf_attn_bwd_bf16 with dq == NULL / dk == dv == NULL issues one of its two launches; train_ops.attn_bwd_bf16(kv_stream=...) puts dK / dV on a
    second stream beside dQ: same bits as the single call, call after call"""
    from viewformer_amd import train_ops as T
    B, H, S = 3, 4, 9
    d, Tn = H * 64, S * 64
    g = torch.Generator().manual_seed(21)
    kv = torch.cuda.Stream(dev)
    for it in range(4):
        qkv = (torch.randn(B * Tn, 3 * d, generator=g) * 0.4).to(dev).to(torch.bfloat16)
        dout = (torch.randn(B * Tn, d, generator=g) * 0.1).to(dev).to(torch.bfloat16)
        q, k, v = qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d]
        o = torch.empty(B * Tn, d, device=dev, dtype=torch.bfloat16)
        lse = T.attn_fwd_lse_bf16(q, k, v, o, B, H, Tn, 64, 3 * d, 3 * d, 3 * d, d, 1.0, -3, drop)
        res = []
        for stream in (None, kv, kv):
            dqkv = torch.full((B * Tn, 3 * d), float('nan'), device=dev, dtype=torch.bfloat16)
            T.attn_bwd_bf16(q, k, v, o, dout, lse, dqkv[:, d:2 * d], dqkv[:, 2 * d:], dqkv[:, :d], B, H, Tn, 64, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d,
                            3 * d, 1.0, -3, drop, kv_stream=stream)
            res.append(dqkv)
        torch.cuda.synchronize()
        assert not torch.isnan(res[0].float()).any()
        for r in res[1:]:
            assert torch.equal(r.view(torch.int16), res[0].view(torch.int16)), it


@pytest.mark.gpu
def test_early_per_layer_optimizer_leaves_the_same_parameters_bit_for_bit(dev):
    """MIGTTrainer.early_optimizer: each layer's AdamWeightDecay update and re-pack on a third stream as soon as its gradients are final.  Three
    steps with it and without it: parameters, both Adam moments and the losses are identical bit for bit (full width, bf16 arm, dropout on)."""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights
    cfg = MIGTConfig(sequence_size=4, n_loss_skip=1, localization_weight='5', pose_multiplier=0.05, dropout=0.1, learning_rate=1e-3, weight_decay=0.05,
                     total_steps=1000, batch_size=2, n_layer=3)
    sd = make_migt_weights(cfg, seed=3)
    g = np.random.Generator(np.random.PCG64(5))
    tokens = torch.from_numpy(g.integers(0, 1024, size=(2, 4, 8, 8)))
    poses = torch.from_numpy(g.standard_normal((2, 4, 7)).astype(np.float32))
    res = []
    for early in (False, True, True):
        tr = MIGTTrainer(MIGT(cfg, precision='bf16').load_state_dict(sd).to(dev))
        tr.early_optimizer = early
        losses = [tr.train_step(poses, tokens)['loss'].clone() for _ in range(3)]
        torch.cuda.synchronize()
        assert (tr._layer_pack_ranges is not None) and tr._pack16 is not None
        res.append((tr.flat_p.clone(), tr.flat_m.clone(), tr.flat_v.clone(), torch.stack(losses)))
    for r in res[1:]:
        for a, b in zip(res[0], r):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.gpu
def test_fused_optimizer_repack_kernel_against_the_two_launches(dev):
    """vf_adamw_flat_pack_f32 on a synthetic flat buffer: weight matrices with both packings, with only one, one inside a no-decay range, small
    tensors between them — parameters, both moments and every packed buffer equal vf_adamw_flat_f32 + vf_gemm_bf16_pack_multi bit for bit."""
    from viewformer_amd import ops, train_ops as T
    g = torch.Generator().manual_seed(11)
    shapes = [(12,), (128, 256), (256,), (128, 128), (4,), (256, 384), (384,), (1024, 128), (8,)]
    offs, n = [], 0
    for s in shapes:
        offs.append(n)
        n += (int(np.prod(s)) + 3) // 4 * 4
    p0 = torch.randn(n, generator=g)
    g0 = torch.randn(n, generator=g) * 0.1
    m0 = torch.randn(n, generator=g) * 0.01
    v0 = torch.rand(n, generator=g) * 0.01
    nodecay = torch.tensor([[offs[2], offs[2] + 256], [offs[5], offs[5] + 256 * 384], [offs[6], offs[6] + 384]], dtype=torch.int64, device=dev)
    want = {1: (True, True), 3: (True, False), 5: (False, True), 7: (True, True)}      # tensor index -> (kn, nk)
    outs = []
    for fused in (False, True):
        p, gr, m, v = (t.clone().to(dev) for t in (p0, g0, m0, v0))
        views = {i: p[offs[i]:offs[i] + int(np.prod(shapes[i]))].view(shapes[i]) for i in want}
        kn = {i: torch.full((int(ops._lib.load().vf_gemm_bf16_packed_elems(*shapes[i])),), float('nan'), dtype=torch.bfloat16, device=dev) for i, w in want.items() if w[0]}
        nk = {i: torch.full((int(ops._lib.load().vf_gemm_bf16_packed_elems(shapes[i][1], shapes[i][0])),), float('nan'), dtype=torch.bfloat16, device=dev)
              for i, w in want.items() if w[1]}
        if fused:
            table = T.adamw_pack_table(p, [(views[i], kn.get(i), nk.get(i)) for i in want], nodecay)
            assert table is not None and table[1] == 4
            T.adamw_flat_pack_(p, gr, m, v, nodecay, 1e-3 * 0.05, 2e-3, 0.9, 0.999, 1e-7, table)
        else:
            T.adamw_flat_(p, gr, m, v, nodecay, 1e-3 * 0.05, 2e-3, 0.9, 0.999, 1e-7)
            ops.pack_bf16_multi([(views[i], False, kn[i]) for i in kn] + [(views[i], True, nk[i]) for i in nk])
        torch.cuda.synchronize()
        outs.append([p, m, v] + [kn[i] for i in sorted(kn)] + [nk[i] for i in sorted(nk)])
    assert not torch.equal(outs[0][0].cpu(), p0)
    for a, b in zip(*outs):
        assert not torch.isnan(a.float()).any()
        assert torch.equal(a.view(torch.int16 if a.dtype == torch.bfloat16 else torch.int32), b.view(torch.int16 if b.dtype == torch.bfloat16 else torch.int32))
    # a table the tile kernel cannot take is refused on the host: rows % 128, overlapping descriptors
    p = p0.clone().to(dev)
    assert T.adamw_pack_table(p, [(p[:64 * 128].view(64, 128), torch.empty(64 * 128, dtype=torch.bfloat16, device=dev), None)]) is None
    w = p[:128 * 128].view(128, 128)
    buf = torch.empty(128 * 128, dtype=torch.bfloat16, device=dev)
    with pytest.raises(Exception):
        T.adamw_pack_table(p, [(w, buf, None), (w, None, buf)])


@pytest.mark.gpu
def test_fused_optimizer_repack_leaves_the_same_trainer_state_bit_for_bit(dev):
    """MIGTTrainer.fused_optimizer_repack: three steps with the optimizer writing the bf16 packings itself and with the separate re-pack launch:
    parameters, moments, losses and every packed operand (layer weights both ways, the tied LM head both ways) identical."""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights
    cfg = MIGTConfig(sequence_size=4, n_loss_skip=1, localization_weight='5', pose_multiplier=0.05, dropout=0.1, learning_rate=1e-3, weight_decay=0.05,
                     total_steps=1000, batch_size=2, n_layer=2)
    sd = make_migt_weights(cfg, seed=3)
    g = np.random.Generator(np.random.PCG64(5))
    tokens = torch.from_numpy(g.integers(0, 1024, size=(2, 4, 8, 8)))
    poses = torch.from_numpy(g.standard_normal((2, 4, 7)).astype(np.float32))
    res = []
    for fused in (False, True):
        tr = MIGTTrainer(MIGT(cfg, precision='bf16').load_state_dict(sd).to(dev))
        tr.fused_optimizer_repack = fused
        losses = [tr.train_step(poses, tokens)['loss'].clone() for _ in range(3)]
        torch.cuda.synchronize()
        assert tr._pack16 is not None and tr._adam_pack is not None and 2 * tr._adam_pack[1] == len(tr._pack16_keep) >= 2 * (4 * cfg.n_layer + 1)
        assert tr._packs_fresh is False
        packs = [out for _, _, out in tr._pack16_keep]               # every layer weight both ways, the tied LM head both ways
        assert {id(x) for x in packs} >= {id(x) for x in tr._lm16} | {id(dn.wp16) for dn in tr.model._dense.values() if dn.wp16 is not None}
        res.append([tr.flat_p.clone(), tr.flat_m.clone(), tr.flat_v.clone(), torch.stack(losses)] + [x.clone() for x in packs])
    assert len(res[0]) == len(res[1])
    for a, b in zip(*res):
        assert torch.equal(a.view(torch.int16 if a.dtype == torch.bfloat16 else torch.int32), b.view(torch.int16 if b.dtype == torch.bfloat16 else torch.int32))


@pytest.mark.gpu
def test_lazy_gradient_zero_gives_the_same_step_and_never_leaves_a_stale_gradient(dev):
    """MIGTTrainer.lazy_gradient_zero: the layers' gradient tensors are not zero-filled, their first writer stores.  Three steps with it and with the
    whole-buffer fill: gradients, parameters, moments and losses equal; the fp32-equivalent arm (accumulating fallback writers) as well; a tensor
    that no kernel wrote is zero-filled when its layer's backward ends, whatever the buffer held."""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights
    cfg = MIGTConfig(sequence_size=4, n_loss_skip=1, localization_weight='5', pose_multiplier=0.05, dropout=0.1, learning_rate=1e-3, weight_decay=0.05,
                     total_steps=1000, batch_size=2, n_layer=2)
    sd = make_migt_weights(cfg, seed=3)
    g = np.random.Generator(np.random.PCG64(5))
    tokens = torch.from_numpy(g.integers(0, 1024, size=(2, 4, 8, 8)))
    poses = torch.from_numpy(g.standard_normal((2, 4, 7)).astype(np.float32))
    for precision in ('bf16', 'f32'):
        res = []
        for lazy in (False, True):
            tr = MIGTTrainer(MIGT(cfg, precision=precision).load_state_dict(sd).to(dev))
            tr.lazy_gradient_zero = lazy
            tr.flat_g.fill_(float('nan'))                              # whatever the buffer held
            losses = [tr.train_step(poses, tokens)['loss'].clone() for _ in range(3)]
            torch.cuda.synchronize()
            assert not tr._unset and torch.isfinite(tr.flat_g).all()
            res.append([tr.flat_g.clone(), tr.flat_p.clone(), tr.flat_m.clone(), tr.flat_v.clone(), torch.stack(losses)])
        for a, b in zip(*res):
            assert torch.equal(a, b)
    tr = MIGTTrainer(MIGT(cfg, precision='bf16').load_state_dict(sd).to(dev))
    tr.flat_g.fill_(7.0)
    tr._begin_gradients()
    a, b = tr.head_range
    assert float(tr.flat_g[a:b].abs().max()) == 0.0 and len(tr._unset) == 6 * cfg.n_layer
    assert tr._first_write('h.1.mlp.c_fc') and not tr._first_write('h.1.mlp.c_fc') and not tr._first_write('ln_f')
    tr._flush_unset('h.1.')
    a1, b1 = tr.layer_ranges[1]
    kept = tr.g('h.1.mlp.c_fc.weight')                                 # its writer was announced: not flushed
    assert float(kept.min()) == 7.0
    z = tr.flat_g[a1:b1].clone()
    lo, hi = tr.slices['h.1.mlp.c_fc.weight'][0] - a1, tr.slices['h.1.mlp.c_fc.bias'][1] - a1
    z[lo:hi] = 0
    assert float(z.abs().max()) == 0.0 and len(tr._unset) == 6
    tr._flush_unset()
    assert not tr._unset and float(tr.flat_g[tr.layer_ranges[0][0]:tr.layer_ranges[0][1]].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('M', [512, 640, 19200])
def test_saved_gelu_derivative_forms_of_the_256_tile_kernel(dev, M):
    """round 6: VF_EPI_GELU_DUAL with gelu_grad writes gelu'(u) (bf16) where the plain form writes u — from the same erf / exp evaluation as the GELU
    beside it, which keeps its bits — and VF_EPI_GELU_BWD with gelu_grad multiplies by that saved derivative.  gelu' against fp64 on the fp32
    pre-activation: one bf16 rounding; the backward product against fp64 of (dy @ W) * gelu'(u): two bf16 roundings; the plain forms refuse the flag."""
    from viewformer_amd import ops, _lib
    K, N = 768, 3072
    g = np.random.Generator(np.random.PCG64(M + 11))
    x16 = torch.from_numpy(g.standard_normal((M, K)).astype(np.float32)).to(dev).to(torch.bfloat16)
    w = torch.from_numpy((g.standard_normal((K, N)) * 0.05).astype(np.float32)).to(dev)
    b = torch.from_numpy(g.standard_normal(N).astype(np.float32)).to(dev)
    wp = ops.pack_dense_kn_bf16(w)
    u32 = torch.empty((M, N), device=dev)
    ops.igemm(x16, wp, M, K, N, u32, bias=b, bf16=True, a16=True)
    u16, f = (torch.empty((M, N), dtype=torch.bfloat16, device=dev) for _ in range(2))
    ops.igemm(x16, wp, M, K, N, u16, bias=b, epilogue=ops.EPI_GELU_DUAL, bf16=True, a16=True, o16=True, out_aux=f)
    gp, f2 = torch.full_like(u16, float('nan')), torch.full_like(f, float('nan'))
    ops.igemm(x16, wp, M, K, N, gp, bias=b, epilogue=ops.EPI_GELU_DUAL, bf16=True, a16=True, o16=True, out_aux=f2, gelu_grad=True)
    assert torch.equal(f2, f)                                                  # the GELU output keeps its bits
    ud = u32.double()
    ref = 0.5 * (1 + torch.erf(ud / 2 ** 0.5)) + ud * torch.exp(-0.5 * ud * ud) / (2 * np.pi) ** 0.5
    assert bool(((gp.double() - ref).abs() <= ref.abs() * 2 ** -8 + 1e-6).all())
    # the backward epilogue on the saved derivative
    dy16 = torch.from_numpy((g.standard_normal((M, N)) * 0.1).astype(np.float32)).to(dev).to(torch.bfloat16)
    wt = ops.pack_dense_nk_bf16(w)                                              # dY @ W^T: [N][K] read as the transposed operand
    du = torch.full((M, K), float('nan'), dtype=torch.bfloat16, device=dev)
    gpk = gp[:, :K].contiguous()                                               # (any [M][K] bf16 table serves as the derivative)
    ops.igemm(dy16, wt, M, N, K, du, res=gpk, epilogue=ops.EPI_GELU_BWD, bf16=True, a16=True, o16=True, res16=True, gelu_grad=True)
    plain = torch.empty((M, K), device=dev)
    ops.igemm(dy16, wt, M, N, K, plain, bf16=True, a16=True)
    assert torch.equal(du, (plain * gpk.float()).to(torch.bfloat16))          # fp32 product of the GEMM's own sums and the loaded value, rounded once
    # old form on the same operands for reference: gelu' evaluated in the epilogue from a bf16 u — the two agree to bf16 precision
    du_old = torch.empty_like(du)
    ops.igemm(dy16, wt, M, N, K, du_old, res=u16[:, :K].contiguous(), epilogue=ops.EPI_GELU_BWD, bf16=True, a16=True, o16=True, res16=True)
    gk = (0.5 * (1 + torch.erf(ud[:, :K] / 2 ** 0.5)) + ud[:, :K] * torch.exp(-0.5 * ud[:, :K] ** 2) / (2 * np.pi) ** 0.5)
    want = plain.double() * gk
    tol = want.abs() * 2 ** -6 + 1e-3 * float(plain.abs().max())
    assert bool(((du.double() - want).abs() <= tol).all()) and bool(((du_old.double() - want).abs() <= tol).all())
    with pytest.raises(_lib.VfError):
        ops.igemm(x16, wp, M, K, N, u32, bias=b, epilogue=ops.EPI_GELU_DUAL, bf16=True, a16=True, out_aux=f, gelu_grad=True)      # fp32 out
    with pytest.raises(_lib.VfError):
        ops.igemm(dy16, wt, M, N, K, du, res=plain, epilogue=ops.EPI_GELU_BWD, bf16=True, a16=True, o16=True, gelu_grad=True)       # fp32 table


@pytest.mark.gpu
def test_saved_gelu_derivative_training_step_stays_within_the_bf16_arm_bound(dev):
    """MIGTTrainer.save_gelu_derivative on / off: the two forms round gelu' at different places (from the fp32 pre-activation once; from the bf16-rounded
    pre-activation in the backward), so gradients agree to the bf16 arm's own precision, not bit for bit; forward outputs (losses of step 1) are identical."""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights
    cfg = MIGTConfig(sequence_size=4, n_loss_skip=1, localization_weight='5', pose_multiplier=0.05, dropout=0.1, learning_rate=1e-3, weight_decay=0.05,
                     total_steps=1000, batch_size=2, n_layer=2)
    sd = make_migt_weights(cfg, seed=3)
    g = np.random.Generator(np.random.PCG64(5))
    tokens = torch.from_numpy(g.integers(0, 1024, size=(2, 4, 8, 8)))
    poses = torch.from_numpy(g.standard_normal((2, 4, 7)).astype(np.float32))
    res = []
    for flag in (False, True):
        tr = MIGTTrainer(MIGT(cfg, precision='bf16').load_state_dict(sd).to(dev))
        tr.save_gelu_derivative = flag
        loss = tr.train_step(poses, tokens, apply_update=False)['loss'].clone()
        torch.cuda.synchronize()
        assert tr._u_is_derivative == flag
        res.append((loss, tr.flat_g.clone()))
    assert torch.equal(res[0][0], res[1][0])
    ga, gb = res[0][1].double(), res[1][1].double()
    for name in ('h.0.mlp.c_fc.weight', 'h.1.mlp.c_fc.weight', 'h.0.attn.c_attn.weight', 'wte.weight'):
        a, b, _ = tr.slices[name]
        assert float((ga[a:b] - gb[a:b]).abs().max() / ga[a:b].abs().max()) < 1.5e-2, name
    assert float((ga - gb).abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize('precision,loc', [('f32', True), ('bf16', True), ('bf16', False)])
def test_last_block_on_the_branch_streams_only(dev, precision, loc):
    """MIGTTrainer.prune_last_block (round 6): forward and backward of the last block's projection / LayerNorm / MLP and of ln_f on the MASK / LOC streams'
    rows only — the main stream's rows of the last block reach no loss.  Without dropout the step equals the all-rows step: losses bit for bit (the rows
    that are kept are the same launches' rows), gradients within fp32 summation order (the weight gradients of the last block add exact zeros for
    the pruned rows in the all-rows step, in another slab partition).  With dropout the two steps draw the last block's masks at different row indices
    (the oracle follows the pruned indexing: test_train_step_with_dropout_...), so only finiteness and determinism are compared here."""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights
    cfg = MIGTConfig(sequence_size=4, n_loss_skip=1, localization_weight='5' if loc else '0', pose_multiplier=0.05, dropout=0.0, learning_rate=1e-3,
                     weight_decay=0.05, total_steps=1000, batch_size=2, n_layer=2)
    sd = make_migt_weights(cfg, seed=3)
    g = np.random.Generator(np.random.PCG64(5))
    tokens = torch.from_numpy(g.integers(0, 1024, size=(2, 4, 8, 8)))
    poses = torch.from_numpy(g.standard_normal((2, 4, 7)).astype(np.float32))
    res = []
    for prune in (False, True, True):
        tr = MIGTTrainer(MIGT(cfg, precision=precision).load_state_dict(sd).to(dev))
        tr.prune_last_block = prune
        met = tr.train_step(poses, tokens, apply_update=False)
        torch.cuda.synchronize()
        res.append((met['loss'].clone(), tr.flat_g.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[1][0], res[2][0])
    assert torch.equal(res[1][1], res[2][1])                                   # deterministic
    ga, gb = res[0][1].double(), res[1][1].double()
    for name in tr.names:
        a, b, _ = tr.slices[name]
        scale = float(ga[a:b].abs().max()) + 1e-30
        assert float((ga[a:b] - gb[a:b]).abs().max()) / scale < (2e-5 if precision == 'f32' else 2e-3), name
    cfg2 = MIGTConfig(**{**cfg.__dict__, 'dropout': 0.1}) if hasattr(cfg, '__dict__') else cfg
    tr = MIGTTrainer(MIGT(cfg2, precision=precision).load_state_dict(sd).to(dev))
    out = [tr.train_step(poses, tokens, apply_update=False)['loss'].clone() for _ in range(1)]
    assert torch.isfinite(out[0]) and torch.isfinite(tr.flat_g).all()
