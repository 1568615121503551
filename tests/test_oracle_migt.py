"""CPU: structural invariants that pin oracle/migt_oracle.py.
The reference transformer is TensorFlow-only and cannot run here: parity unpinned,
these are the known-answer properties derivable from branching_attention.py / migt.py."""
import numpy as np
import pytest
import torch

from conftest import TINY_MIGT
from oracle import migt_oracle as mg
from viewformer_amd.config import MIGTConfig
from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch


def _qkv(b=2, h=2, s=4, l=4, d=8, seed=0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(b, h, s, l, d, generator=g, dtype=dtype) * 2 for _ in range(3)]


def _dense_loops(k, v, q):
    """independent restatement: per query view, softmax over the explicitly masked key set."""
    b, h, s, l, d = k.shape
    out = torch.zeros_like(q)
    kf, vf = k.reshape(b, h, s * l, d), v.reshape(b, h, s * l, d)
    for i in range(s):
        w = q[:, :, i] @ kf.transpose(-1, -2)                      # [b,h,l,s*l]
        w[..., (i + 1) * l:] = -1e4
        p = torch.softmax(w, -1)
        out[:, :, i] = p @ vf
    return out


def test_single_stream_is_block_causal_dense_attention():
    k, v, q = _qkv()
    a = mg.compute_causal_block_multiend_attention([k], [v], [q])
    assert len(a) == 1
    assert torch.allclose(a[0], _dense_loops(k, v, q), atol=1e-12)
    # masked keys really get zero weight: perturbing a future view leaves earlier outputs untouched
    k2, v2 = k.clone(), v.clone()
    k2[:, :, -1] += 3
    v2[:, :, -1] -= 5
    a2 = mg.compute_causal_block_multiend_attention([k2], [v2], [q])[0]
    assert torch.equal(a2[:, :, :-1], a[0][:, :, :-1])


def test_branch_equals_attention_over_past_main_views_plus_own_tokens():
    k, v, q = _qkv(seed=1)
    kb, vb, qb = _qkv(seed=2)
    outs = mg.compute_causal_block_multiend_attention([k, kb], [v, vb], [q, qb])
    b, h, s, l, d = k.shape
    for i in range(s):
        keys = torch.cat([k[:, :, :i].reshape(b, h, i * l, d), kb[:, :, i]], 2)
        vals = torch.cat([v[:, :, :i].reshape(b, h, i * l, d), vb[:, :, i]], 2)
        ref = torch.softmax(qb[:, :, i] @ keys.transpose(-1, -2), -1) @ vals
        assert torch.allclose(outs[1][:, :, i], ref, atol=1e-12)


def _setup(loc, seed=0, S=4, B=2):
    cfg = MIGTConfig(**TINY_MIGT, localization_weight='1' if loc else '0', pose_multiplier=0.2)
    sd = make_migt_weights(cfg, seed=seed, std=0.08)
    g = np.random.Generator(np.random.PCG64(seed + 11))
    t = cfg.token_image_size
    ids = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(B, S, t, t)))
    _, cams = synthetic_scene_batch(B, S, 8, seed)
    return cfg, sd, ids, torch.from_numpy(cams)


def test_train_graph_stream1_equals_masked_inference():
    """§8c(iii): MASK-stream logits of view i in the multi-stream graph == last-view
    logits of a single-stream call on [views 0..i-1, MASK] (migt.py:392-396 vs
    evaluate_transformer.py:120-122)."""
    for loc in (False, True):
        cfg, sd, ids, cams = _setup(loc)
        full = mg.migt_forward(sd, cfg, ids, cams, dtype=torch.float64, compute_losses=True)['logits']
        for i in range(ids.shape[1]):
            inp = torch.cat([ids[:, :i], torch.full_like(ids[:, :1], cfg.n_embeddings)], 1)
            one = mg.migt_forward(sd, cfg, inp, cams[:, :i + 1], dtype=torch.float64)['logits']
            assert torch.allclose(one[:, -1], full[:, i], atol=1e-9), (loc, i)


def test_localization_stream_equals_localization_inference():
    """3rd stream (tokens + LOC) of the training graph == the evaluator's second pass
    (evaluate_transformer.py:134-136; migt.py:387-390,398-401)."""
    cfg, sd, ids, cams = _setup(True, seed=3)
    full = mg.migt_forward(sd, cfg, ids, cams, dtype=torch.float64, compute_losses=True)['pose_prediction']
    for i in range(1, ids.shape[1]):
        one = mg.migt_forward(sd, cfg, ids[:, :i + 1], cams[:, :i], dtype=torch.float64)['pose_prediction']
        assert torch.allclose(one[:, -1], full[:, i], atol=1e-9), i


def test_fp32_arm_close_to_fp64_arm():
    cfg, sd, ids, cams = _setup(True, seed=5)
    a = mg.migt_forward(sd, cfg, ids, cams[:, :-1], dtype=torch.float32)
    b = mg.migt_forward(sd, cfg, ids, cams[:, :-1], dtype=torch.float64)
    assert a['logits'].dtype == torch.float32
    assert (a['logits'].double() - b['logits']).abs().max() < 1e-4
    assert (a['pose_prediction'].double() - b['pose_prediction']).abs().max() < 1e-4


def test_vqk_split_order_and_no_scale_matter():
    """Guards the two classic mistakes (Appendix C): (V,Q,K) thirds and no 1/sqrt(d)."""
    cfg, sd, ids, cams = _setup(False, seed=7)
    base = mg.migt_forward(sd, cfg, ids, cams, dtype=torch.float64)['logits']
    sd2 = dict(sd)
    w = sd['h.0.attn.c_attn.weight']
    d = cfg.d_model
    sd2['h.0.attn.c_attn.weight'] = np.concatenate([w[:, d:2 * d], w[:, 2 * d:], w[:, :d]], 1)   # as if (Q,K,V)
    other = mg.migt_forward(sd2, cfg, ids, cams, dtype=torch.float64)['logits']
    assert (base - other).abs().max() > 1e-3


def test_relative_cameras_round_trip_and_reduce():
    _, cams = synthetic_scene_batch(3, 5, 8, 2)
    c = torch.from_numpy(cams).double()
    rel, tr = mg.to_relative_cameras(c)
    assert torch.allclose(rel[:, 0, :3], torch.zeros(3, 3, dtype=torch.float64), atol=1e-12)
    assert torch.allclose(rel[:, 0, 3:].abs(), torch.tensor([1., 0, 0, 0], dtype=torch.float64).expand(3, 4), atol=1e-9)
    back = mg.from_relative_cameras(rel, tr)
    assert torch.allclose(back[..., :3], c[..., :3], atol=1e-9)
    # quaternions equal up to sign
    dot = (back[..., 3:] * c[..., 3:]).sum(-1).abs()
    assert torch.allclose(dot, torch.ones_like(dot), atol=1e-9)
    n = mg.normalize_cameras(rel)
    assert (n[..., 3] >= 0).all() and torch.allclose(n[..., 3:].norm(dim=-1), torch.ones(3, 5, dtype=torch.float64))
    r = mg.reduce_cameras(n.unsqueeze(2).expand(3, 5, 16, 7), -2)
    assert torch.allclose(r, n, atol=1e-9)


# ------------------------------------------------------------------------------------------------------------------------
# the INDEPENDENT fp64 numpy transliteration (oracle/migt_numpy64.py: no code shared with migt_oracle.py,
# written op for op from migt.py:338-455 + branching_attention.py:82-126) against the torch restatement the GPU tests use.
def _np_inputs(ids, cams):
    return ids.numpy(), cams.numpy().astype(np.float32)


def test_independent_numpy64_transliteration_agrees_on_inference_graphs():
    from oracle import migt_numpy64 as n64
    for loc in (False, True):
        cfg, sd, ids, cams = _setup(loc, seed=4, S=5)
        i_np, c_np = _np_inputs(ids, cams)
        # generation pass: all S poses
        a = mg.migt_forward(sd, cfg, ids, cams, dtype=torch.float64)
        b = n64.migt_call(sd, cfg, dict(input_ids=i_np, poses=c_np))
        assert np.abs(a['logits'].numpy() - b['logits']).max() < 1e-10
        assert np.abs(a['hidden_states'][0].numpy() - b['hidden_states'][0].reshape(a['hidden_states'][0].shape)).max() < 1e-10
        if loc:
            # localization pass: S - 1 poses, LOC embedding on the last view (migt.py:387-390)
            a = mg.migt_forward(sd, cfg, ids, cams[:, :-1], dtype=torch.float64)
            b = n64.migt_call(sd, cfg, dict(input_ids=i_np, poses=c_np[:, :-1]))
            assert np.abs(a['pose_prediction'].numpy() - b['pose_prediction']).max() < 1e-10
            assert np.abs(a['logits'].numpy() - b['logits']).max() < 1e-10
            # multi-context evaluator graph: MASK stream + LOC stream (evaluate_transformer_multictx.py:60-73)
            q_cam = cams[:, -1:].expand(-1, ids.shape[1], -1).contiguous()
            q_tok = ids[:, -1:].expand(-1, ids.shape[1], -1, -1).contiguous()
            a = mg.migt_forward(sd, cfg, ids, cams, localization_tokens=q_tok, output_poses=q_cam, dtype=torch.float64)
            b = n64.migt_call(sd, cfg, dict(input_ids=i_np, poses=c_np, localization_tokens=q_tok.numpy(), output_poses=q_cam.numpy()))
            assert len(b['hidden_states']) == 3
            assert np.abs(a['logits'].numpy() - b['logits']).max() < 1e-10
            assert np.abs(a['pose_prediction'].numpy() - b['pose_prediction']).max() < 1e-10


@pytest.mark.parametrize('opts', [dict(), dict(label_smoothing=0.1, image_generation_weight=0.7),
                                  dict(use_dynamic_pose_loss=True, localization_weight='cosine(0,2,10)'),
                                  dict(localization_weight='0')])
def test_independent_numpy64_transliteration_agrees_on_training_losses(opts):
    """the 2-/3-stream training graph and every loss term (migt.py:416-448) — two restatements, one number"""
    from oracle import migt_numpy64 as n64
    from oracle import train_oracle as to
    kw = dict(TINY_MIGT, pose_multiplier=0.2, n_loss_skip=1, localization_weight='1')
    kw.update(opts)
    cfg = MIGTConfig(**kw)
    sd = make_migt_weights(cfg, seed=9, std=0.08)
    g = np.random.Generator(np.random.PCG64(5))
    ids = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(3, 4, 4, 4)))
    _, cams = synthetic_scene_batch(3, 4, 8, 6)
    cams = torch.from_numpy(cams)
    rpm = np.array([1.3, 0.6, 1.0])
    sd_t = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    total, met = to.losses(sd_t, cfg, cams, ids, step=3, pose_factors=torch.from_numpy(rpm))
    b = n64.migt_call(sd, cfg, dict(input_ids=ids.numpy(), poses=cams.numpy()), compute_losses=True, train_counter=3,
                      random_pose_multiplier=rpm)
    assert abs(float(total) - float(np.mean(b['loss']))) < 1e-10 * max(1.0, abs(float(total)))      # reduce_mean(loss), migt.py:476
    assert abs(float(met['ce_loss']) - float(b['ce_loss'].mean())) < 1e-10
    if cfg.use_localization:
        assert abs(float(met['pose_pos_loss']) - float(b['pose_pos_loss'].mean())) < 1e-10
        assert abs(float(met['pose_ori_loss']) - float(b['pose_ori_loss'].mean())) < 1e-10
        assert abs(met['localization_weight'] - b['localization_weight']) < 1e-12
    else:
        assert 'pose_prediction' not in b


def test_independent_numpy64_camera_frames_agree():
    from oracle import migt_numpy64 as n64
    _, cams = synthetic_scene_batch(3, 5, 8, 2)
    c = torch.from_numpy(cams).double()
    rel, tr = mg.to_relative_cameras(c)
    rel2, tr2 = n64.to_relative_cameras(cams.astype(np.float64))
    assert np.abs(rel.numpy() - rel2).max() < 1e-12 and np.abs(tr.numpy() - tr2).max() == 0
    assert np.abs(mg.normalize_cameras(rel).numpy() - n64.normalize_cameras(rel2)).max() < 1e-12
    x = torch.randn(2, 3, 16, 7, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    assert np.abs(mg.reduce_cameras(x, -2).numpy() - n64.reduce_cameras(x.numpy(), -2)).max() < 1e-12


# ---------------------------------------------------------------------------------------------- third-party anchor (Hugging Face GPT-2)
def _hf():
    from conftest import load_golden
    return load_golden('migt_hf_gpt2.npz')


def test_oracle_matches_hugging_face_gpt2_on_migt_weights():
    """tests/golden/migt_hf_gpt2.npz holds outputs of ``transformers``' GPT-2 (an independent, third-party implementation of the architecture
    migt.py is built from) run on MIGT weights with the reference's three deviations expressed through GPT-2's own config / inputs — no
    score scaling, (V, Q, K) split, block-causal 0 / -1e4 mask — and the pose MLPs evaluated with transformers' Conv1D
    (tests/golden/make_hf_gpt2_golden.py).  The oracle reproduces them to fp32 storage precision: single-stream logits of every view, the
    multi-context pass's MASK-stream logits and LOC-stream camera predictions (one GPT-2 call per view there: the branch semantics of
    branching_attention.py:82-126), the relative-camera transform, and the full-size 12-layer model's last-view logits.  (Not the reference
    itself: the MIGT oracle stays formally 'parity unpinned' — this replaces agreement between two restatements by one author with
    agreement with somebody else's code.)"""
    g = _hf()
    cfg = MIGTConfig(n_embeddings=64, n_head=2, d_model=128, n_layer=2, token_image_size=4, sequence_size=4, localization_weight='1', pose_multiplier=0.2)
    sd = make_migt_weights(cfg, seed=int(g['tiny_seed']), std=float(g['tiny_std']))
    ids = torch.from_numpy(g['tiny_ids'])
    cams = mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(g['tiny_cams_raw']).double())[0])
    assert (cams - torch.from_numpy(g['tiny_cams']).double()).abs().max() < 1e-6          # camera frame change vs plain quaternion algebra
    cams = cams.float()
    out = mg.migt_forward(sd, cfg, ids, cams, dtype=torch.float64)
    e1 = (out['logits'] - torch.from_numpy(g['tiny_logits']).double()).abs().max().item()
    in_ids = torch.cat([ids[:, :-1], torch.full_like(ids[:, :1], cfg.n_embeddings)], 1)
    ctx = torch.cat([cams[:, :-1], torch.zeros_like(cams[:, :1])], 1)
    S = ids.shape[1]
    multi = mg.migt_forward(sd, cfg, in_ids, ctx, localization_tokens=ids[:, -1:].expand(-1, S, -1, -1), output_poses=cams[:, -1:].expand(-1, S, -1),
                            dtype=torch.float64)
    e2 = (multi['logits'] - torch.from_numpy(g['tiny_multi_logits']).double()).abs().max().item()
    e3 = (multi['pose_prediction'] - torch.from_numpy(g['tiny_multi_pose']).double()).abs().max().item()
    scale = float(np.abs(g['tiny_logits']).max())
    print(f'oracle vs HF GPT-2 (tiny): logits {e1:.2e}, multi-stream logits {e2:.2e}, pose prediction {e3:.2e} (|logit| max {scale:.2f})')
    assert e1 < 2e-6 * scale and e2 < 2e-6 * scale and e3 < 2e-6 * max(1.0, float(np.abs(g['tiny_multi_pose']).max()))


def test_oracle_full_size_matches_hugging_face_gpt2():
    """the bench's transformer (12 layers, d = 768, 12 heads, 6 context views + the MASK view): last-view logits of the fp64 oracle against
    the fp64 GPT-2 run stored as fp32"""
    g = _hf()
    cfg = MIGTConfig(sequence_size=6, n_loss_skip=1, pose_multiplier=0.2, localization_weight='cosine(0,1,120000)')
    sd = make_migt_weights(cfg, seed=int(g['full_seed']))
    out = mg.migt_forward(sd, cfg, torch.from_numpy(g['full_ids']), torch.from_numpy(g['full_cams']), dtype=torch.float64)['logits'][:, -1]
    ref = torch.from_numpy(g['full_logits_last']).double()
    err = (out - ref).abs().max().item()
    print(f'oracle vs HF GPT-2 (full size): {err:.2e} of |logit| max {ref.abs().max().item():.3f}')
    assert err < 2e-6 * max(1.0, ref.abs().max().item())
    # the evaluator's localization pass (evaluate_transformer.py:134-140): real codes everywhere, cameras of the context views only, LOC embedding
    # on the last view, camera head on its 64 tokens
    loc = mg.migt_forward(sd, cfg, torch.from_numpy(g['full_loc_codes']), torch.from_numpy(g['full_cams'])[:, :-1], dtype=torch.float64)
    e_loc = (loc['pose_prediction'][:, -1] - torch.from_numpy(g['full_loc_pose_last']).double()).abs().max().item()
    print(f'oracle vs HF GPT-2 (full size, localization pass): camera head {e_loc:.2e}')
    assert e_loc < 2e-6


def test_training_losses_and_gradients_match_hugging_face_gpt2_autograd():
    """the TRAINING graph (MIGT.train_step: main + MASK + LOC stream, token cross-entropy and pose MSE over views >= n_loss_skip, batch mean) built
    on ``transformers``' GPT-2 with torch autograd (tests/golden/make_hf_gpt2_golden.py::train_graph: one GPT-2 call per view and branch, losses
    written out there independently of oracle/): the loss terms and the gradient of EVERY variable — L2 norm and 48 sampled entries per tensor,
    GPT-2's c_attn gradient permuted back to (V, Q, K) — against fp64 autograd over oracle/train_oracle.py.  The HIP training step is tested
    against that oracle (tests/test_train.py), so its forward, losses and backward now hang on a third-party implementation too."""
    from oracle import train_oracle as to
    g = _hf()
    cfg = MIGTConfig(n_embeddings=64, n_head=2, d_model=128, n_layer=2, token_image_size=4, sequence_size=4, localization_weight='2', pose_multiplier=0.2,
                     n_loss_skip=1, dropout=0.0)
    sd = make_migt_weights(cfg, seed=int(g['tiny_seed']), std=float(g['tiny_std']))
    grads, met = to.gradients(sd, cfg, torch.from_numpy(g['tiny_cams']), torch.from_numpy(g['tiny_ids']), step=0)
    assert abs(met['loss'] - float(g['train_loss'])) < 1e-6 and abs(met['ce_loss'] - float(g['train_ce'])) < 1e-6
    assert abs(met['pose_pos_loss'] - float(g['train_pos'])) < 1e-6 and abs(met['pose_ori_loss'] - float(g['train_ori'])) < 1e-6
    names = [str(n) for n in g['train_names']]
    assert set(names) == {k for k in sd if k != 'pose_loss_weighting_criterion.pos_ori_weights'}
    worst = 0.0
    for i, n in enumerate(names):
        ref_norm = float(g['train_norms'][i])
        got = grads[n].reshape(-1).double()
        samp = got[torch.from_numpy(g['train_idx'][i])]
        e = max(abs(float(got.norm()) - ref_norm), float((samp - torch.from_numpy(g['train_samples'][i]).double()).abs().max()))
        worst = max(worst, e / max(ref_norm, 1e-12))
        assert e <= 2e-6 * max(ref_norm, 1e-9) + 1e-12, (n, e, ref_norm)
    print(f'oracle autograd vs HF GPT-2 autograd: loss {met["loss"]:.6f} == {float(g["train_loss"]):.6f}; worst gradient deviation {worst:.2e} of the tensor norm over {len(names)} variables')


def test_relative_camera_frame_change_matches_scipy_rotations():
    """evaluate_transformer.py:70-94 / geometry_tf.py:6-13,53-68 restated in oracle/ and in viewformer_amd/geometry.py against a third-party
    implementation: scipy.spatial.transform.Rotation (positions rotated by the inverse of the first view's rotation after subtracting its
    position; orientations composed with that inverse; unit quaternion with w >= 0).  Quaternions are (w, x, y, z) in the reference, (x, y, z, w)
    in scipy."""
    from scipy.spatial.transform import Rotation as R
    from viewformer_amd import geometry
    _, cams = synthetic_scene_batch(3, 6, 8, 17)
    cams = torch.from_numpy(cams).double()
    rel, _ = mg.to_relative_cameras(cams)
    rel = mg.normalize_cameras(rel)
    rel_p = geometry.normalize_cameras(geometry.to_relative_cameras(cams.float())[0]).double()
    for b in range(cams.shape[0]):
        q = cams[b, :, 3:].numpy()
        rot = R.from_quat(np.concatenate([q[:, 1:], q[:, :1]], 1))             # scipy: scalar last
        r0inv = rot[0].inv()
        pos = r0inv.apply((cams[b, :, :3] - cams[b, :1, :3]).numpy())
        qq = (r0inv * rot).as_quat()
        qq = np.concatenate([qq[:, 3:], qq[:, :3]], 1)
        qq = qq * np.where(qq[:, :1] >= 0, 1.0, -1.0)
        want = torch.from_numpy(np.concatenate([pos, qq], 1))
        # (the inputs' quaternions are unit to fp32 precision only: the reference rotates by the CONJUGATE, scipy by the normalised inverse: 1e-7)
        assert (rel[b] - want).abs().max() < 1e-6, (rel[b] - want).abs().max()
        assert (rel_p[b] - want).abs().max() < 3e-6
