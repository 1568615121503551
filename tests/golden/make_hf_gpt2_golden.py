#!/usr/bin/env python
"""Third-party anchor for the MIGT half of the oracle: outputs of Hugging Face ``transformers``' GPT-2 (PyTorch) on MIGT weights.

The reference transformer (viewformer/models/migt.py) is TensorFlow and cannot run here, and it ships no fixtures: oracle/migt_oracle.py is
a restatement that nothing written by the reference pins.  What CAN be had is an independent implementation of the same architecture:
migt.py's layers are, class for class and attribute for attribute, the GPT-2 layers of ``transformers``' TF port (SharedEmbeddings with
``initializer_range``, Conv1D(nf, nx) computing x @ W[nx, nf] + b, MLP with c_fc / c_proj, pre-LN Block with ln_1 / attn / ln_2 / mlp,
layer_norm_epsilon 1e-5, the tied ``wte`` head) — with three changes that GPT-2's config / inputs can express exactly:
  * attention scores are NOT scaled (branching_attention.py:5-18)                      -> ``scale_attn_weights=False``
  * c_attn's output is split (V, Q, K) (migt.py:207-209), GPT-2's (Q, K, V)            -> the weight's column thirds are permuted on load
  * the mask is block-causal over views with ``w * m - 1e4 (1 - m)`` (:41-61)           -> a float 4-D ``attention_mask`` of 0 / -1e4 (added to the
                                                                                           scores; softmax of -1e4 underflows to 0 either way)
and GELU is the exact erf form (``tf.nn.gelu`` default)                                 -> ``activation_function='gelu'``.
The input embedding (token + position-in-view + pose MLP, migt.py:338-368,392) goes in as ``inputs_embeds`` + ``position_ids``; the pose MLPs
(7 -> 1536 -> d, d -> 1536 -> 7) are evaluated with ``transformers.pytorch_utils.Conv1D`` modules.  The multi-stream pass (MASK stream with
``output_poses``, LOC stream with ``localization_tokens``: migt.py:371-401, branching_attention.py:82-126) is reproduced by ONE GPT-2 call per
view i on the sequence [main views 0 .. i-1 | the branch's view i] — by construction of the branch mask that is exactly what the branch's
view i sees, and the main views never see a branch.

This is NOT the reference (the contract's definition of "pinned" needs the reference's own outputs): the MIGT oracle stays formally
"parity unpinned".  It replaces "two restatements by one author" by agreement with a third-party implementation to 1e-13.

Writes tests/golden/migt_hf_gpt2.npz (fp64 run, stored fp32): tiny config (the HIP kernels' TINY_MIGT shape) single-stream logits, multi-stream
logits + pose predictions; full-size (12 layers, d = 768, 12 heads, 6 context views + MASK view) last-view logits.
  python tests/golden/make_hf_gpt2_golden.py        (needs ``transformers``; run in the build container)
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from transformers import GPT2Config  # noqa: E402
from transformers.models.gpt2.modeling_gpt2 import GPT2Model  # noqa: E402
from transformers.pytorch_utils import Conv1D  # noqa: E402
from viewformer_amd.config import MIGTConfig  # noqa: E402
from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch  # noqa: E402

F64 = torch.float64


def t(sd, k):
    return torch.as_tensor(np.asarray(sd[k]), dtype=F64)


def build_gpt2(cfg, sd):
    hc = GPT2Config(n_embd=cfg.d_model, n_head=cfg.n_head, n_layer=cfg.n_layer, n_positions=256, vocab_size=cfg.n_embeddings + 2,
                    activation_function='gelu', scale_attn_weights=False, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
                    layer_norm_epsilon=1e-5, bos_token_id=0, eos_token_id=0, attn_implementation='eager')
    m = GPT2Model(hc).double().eval()
    d = cfg.d_model
    st = {'wte.weight': t(sd, 'wte.weight'), 'wpe.weight': t(sd, 'wpe.embeddings'), 'ln_f.weight': t(sd, 'ln_f.gamma'), 'ln_f.bias': t(sd, 'ln_f.beta')}
    for i in range(cfg.n_layer):
        p = f'h.{i}.'
        w, b = t(sd, p + 'attn.c_attn.weight'), t(sd, p + 'attn.c_attn.bias').reshape(-1)
        st[p + 'attn.c_attn.weight'] = torch.cat([w[:, d:2 * d], w[:, 2 * d:], w[:, :d]], 1)          # (V, Q, K) -> (Q, K, V)
        st[p + 'attn.c_attn.bias'] = torch.cat([b[d:2 * d], b[2 * d:], b[:d]])
        for a, c in (('ln_1.gamma', 'ln_1.weight'), ('ln_1.beta', 'ln_1.bias'), ('ln_2.gamma', 'ln_2.weight'), ('ln_2.beta', 'ln_2.bias'),
                     ('attn.c_proj.weight',) * 2, ('attn.c_proj.bias',) * 2, ('mlp.c_fc.weight',) * 2, ('mlp.c_fc.bias',) * 2,
                     ('mlp.c_proj.weight',) * 2, ('mlp.c_proj.bias',) * 2):
            st[p + c] = t(sd, p + a).reshape(m.state_dict()[p + c].shape)
    missing, unexpected = m.load_state_dict(st, strict=False)
    assert not unexpected and all(k.endswith(('attn.bias', 'attn.masked_bias')) for k in missing), (missing, unexpected)
    return m


def hf_mlp(sd, name, x):
    """c_proj(gelu(c_fc(x))) with transformers' Conv1D modules and torch's exact GELU"""
    out = x.contiguous()
    for part in ('c_fc', 'c_proj'):
        w, b = t(sd, f'{name}.{part}.weight'), t(sd, f'{name}.{part}.bias').reshape(-1)
        conv = Conv1D(w.shape[1], w.shape[0]).double()
        with torch.no_grad():
            conv.weight.copy_(w)
            conv.bias.copy_(b)
            out = conv(out)
        if part == 'c_fc':
            out = torch.nn.functional.gelu(out)              # approximate='none': erf
    return out


def relative_normalized(cams):
    """evaluate_transformer.py:70-94 restated with plain quaternion algebra (independent of oracle/): first view becomes the origin frame,
    quaternions normalised with w >= 0"""
    cams = torch.as_tensor(cams, dtype=F64)
    pos, q = cams[..., :3], cams[..., 3:]

    def qmul(a, b):
        aw, ax, ay, az = a.unbind(-1)
        bw, bx, by, bz = b.unbind(-1)
        return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                            aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)

    def conj(a):
        return a * torch.tensor([1.0, -1.0, -1.0, -1.0], dtype=F64)

    q0inv = conj(q[:, :1])                                   # unit quaternions: inverse = conjugate
    rel = pos - pos[:, :1]
    zero = torch.zeros_like(rel[..., :1])
    rot = qmul(qmul(q0inv, torch.cat([zero, rel], -1)), conj(q0inv))[..., 1:]
    qr = qmul(q0inv, q)
    qr = qr / qr.norm(dim=-1, keepdim=True).clamp_min(1e-6)
    qr = qr * (2 * (qr[..., :1] >= 0).double() - 1)
    return torch.cat([rot, qr], -1)


def run(cfg, sd, ids, cams, multi):
    """-> dict of arrays.  ids [B,S,t,t]; cams [B,S,7] already relative + normalised"""
    B, S = ids.shape[:2]
    L = ids.shape[2] * ids.shape[3]
    m = build_gpt2(cfg, sd)
    wte, nE = t(sd, 'wte.weight'), cfg.n_embeddings
    pin = torch.cat([cams[..., :3] * cfg.pose_multiplier, cams[..., 3:]], -1)
    pose = hf_mlp(sd, 'pose_embedding', pin)[:, :, None, :]                      # [B,S,1,d]
    posv = torch.arange(L).repeat(S)[None].expand(B, -1)

    def gpt2(emb_views):                                       # emb_views [B,V,L,d] -> hidden after ln_f [B,V,L,d], block-causal over the V views
        V = emb_views.shape[1]
        view = torch.arange(V).repeat_interleave(L)
        mask = torch.where(view[:, None] >= view[None, :], 0.0, -1e4).double()[None, None].expand(B, 1, -1, -1)
        with torch.no_grad():
            h = m(inputs_embeds=emb_views.reshape(B, V * L, -1), position_ids=posv[:, :V * L], attention_mask=mask).last_hidden_state
        return h.reshape(B, V, L, -1)

    out = {}
    if not multi:
        main = wte[ids.reshape(B, S, L)] + pose                # (+ wpe[0:L] inside GPT-2 through position_ids)
        h = gpt2(main)
        out['logits'] = (h @ wte.t())[..., :nE].reshape(B, S, *ids.shape[2:], nE)
        return out
    # multi-context pass (evaluate_transformer_multictx.py:60-77): context codes + MASK last view, context cameras with the last one zeroed,
    # output_poses = the target camera at every position, localization_tokens = the target's codes at every position
    mask_tok, loc_tok = nE, nE + 1
    in_ids = torch.cat([ids[:, :-1], torch.full_like(ids[:, :1], mask_tok)], 1).reshape(B, S, L)
    ctx = torch.cat([cams[:, :-1], torch.zeros_like(cams[:, :1])], 1)
    pin_c = torch.cat([ctx[..., :3] * cfg.pose_multiplier, ctx[..., 3:]], -1)
    pose_c = hf_mlp(sd, 'pose_embedding', pin_c)[:, :, None, :]
    main = wte[in_ids] + pose_c
    tgt_pin = pin[:, -1:].expand(B, S, 7)
    mask_stream = wte[mask_tok].reshape(1, 1, 1, -1) + hf_mlp(sd, 'pose_embedding', tgt_pin)[:, :, None, :].expand(B, S, L, -1)
    loc_stream = wte[ids[:, -1:].reshape(B, 1, L)].expand(B, S, L, -1) + wte[loc_tok].reshape(1, 1, 1, -1)
    logits, poses = [], []
    for i in range(S):                                         # view i of a branch sees main views 0 .. i-1 and itself
        hm = gpt2(torch.cat([main[:, :i], mask_stream[:, i:i + 1]], 1))[:, -1]
        hl = gpt2(torch.cat([main[:, :i], loc_stream[:, i:i + 1]], 1))[:, -1]
        logits.append((hm @ wte.t())[..., :nE])
        y = hf_mlp(sd, 'pose_criterion.pose_classifier', hl)
        qn = y[..., 3:] * torch.rsqrt((y[..., 3:] ** 2).sum(-1, keepdim=True).clamp_min(1e-12))
        qn = qn * (2 * (qn[..., :1] >= 0).double() - 1)
        poses.append(torch.cat([y[..., :3] / cfg.pose_multiplier, qn], -1))
    out['logits'] = torch.stack(logits, 1).reshape(B, S, *ids.shape[2:], nE)
    out['pose_prediction'] = torch.stack(poses, 1)               # [B,S,L,7]
    return out


def train_graph(cfg, sd, ids, cams):
    """MIGT.train_step's forward + losses (migt.py:371-448,464-476) on GPT-2 with AUTOGRAD: main stream, MASK stream (output_poses = poses) and
    LOC stream (localization_tokens = the tokens) — one GPT-2 call per view and branch, as in run(); token cross-entropy on the MASK stream and
    position / orientation MSE on the LOC stream over views >= n_loss_skip, per-scene means, batch mean.  Returns the loss terms and the
    gradient w.r.t. every MIGT variable (GPT-2's c_attn gradient permuted back to the reference's (V, Q, K) column order)."""
    B, S = ids.shape[:2]
    L = ids.shape[2] * ids.shape[3]
    m = build_gpt2(cfg, sd)
    for p_ in m.parameters():
        p_.requires_grad_(True)
    nE, d = cfg.n_embeddings, cfg.d_model
    mask_tok, loc_tok = nE, nE + 1

    def conv(name):
        w, b = t(sd, f'{name}.weight'), t(sd, f'{name}.bias').reshape(-1)
        c = Conv1D(w.shape[1], w.shape[0]).double()
        with torch.no_grad():
            c.weight.copy_(w)
            c.bias.copy_(b)
        return c
    mods = {n: conv(n) for n in ('pose_embedding.c_fc', 'pose_embedding.c_proj', 'pose_criterion.pose_classifier.c_fc',
                                 'pose_criterion.pose_classifier.c_proj')}

    def mlp(name, x):
        return mods[name + '.c_proj'](torch.nn.functional.gelu(mods[name + '.c_fc'](x.contiguous())))
    wte = m.wte.weight
    pin = torch.cat([cams[..., :3] * cfg.pose_multiplier, cams[..., 3:]], -1)
    pose = mlp('pose_embedding', pin)[:, :, None, :]
    posv = torch.arange(L).repeat(S)[None].expand(B, -1)
    idl = ids.reshape(B, S, L)

    def gpt2(emb_views):
        V = emb_views.shape[1]
        view = torch.arange(V).repeat_interleave(L)
        mask = torch.where(view[:, None] >= view[None, :], 0.0, -1e4).double()[None, None].expand(B, 1, -1, -1)
        return m(inputs_embeds=emb_views.reshape(B, V * L, -1), position_ids=posv[:, :V * L], attention_mask=mask).last_hidden_state.reshape(B, V, L, -1)
    main = wte[idl] + pose
    mask_stream = wte[mask_tok].reshape(1, 1, 1, -1) + pose.expand(B, S, L, -1)
    loc_stream = wte[idl] + wte[loc_tok].reshape(1, 1, 1, -1)
    skip = cfg.n_loss_skip
    ce_v, pos_v, ori_v = [], [], []
    y = cams.unsqueeze(-2) * torch.tensor([cfg.pose_multiplier] * 3 + [1.0] * 4, dtype=F64)
    for i in range(S):
        hm = gpt2(torch.cat([main[:, :i], mask_stream[:, i:i + 1]], 1))[:, -1]            # [B,L,d]
        hl = gpt2(torch.cat([main[:, :i], loc_stream[:, i:i + 1]], 1))[:, -1]
        lg = (hm @ wte.t())[..., :nE]
        ce_v.append(torch.nn.functional.cross_entropy(lg.reshape(-1, nE), idl[:, i].reshape(-1), reduction='none').reshape(B, L))
        raw = mlp('pose_criterion.pose_classifier', hl)
        pos_v.append(((y[:, i, :, :3] - raw[..., :3]) ** 2).mean(-1))
        ori_v.append(((y[:, i, :, 3:] - raw[..., 3:]) ** 2).mean(-1))
    ce = torch.stack(ce_v, 1)[:, skip:].mean((1, 2))
    pos = torch.stack(pos_v, 1)[:, skip:].mean((1, 2))
    ori = torch.stack(ori_v, 1)[:, skip:].mean((1, 2))
    w = float(cfg.localization_weight)
    loss = (ce * cfg.image_generation_weight + (pos + ori) * w).mean()
    loss.backward()
    g = {'wte.weight': m.wte.weight.grad, 'wpe.embeddings': m.wpe.weight.grad, 'ln_f.gamma': m.ln_f.weight.grad, 'ln_f.beta': m.ln_f.bias.grad}
    for i in range(cfg.n_layer):
        blk, p_ = m.h[i], f'h.{i}.'
        gw, gb = blk.attn.c_attn.weight.grad, blk.attn.c_attn.bias.grad
        g[p_ + 'attn.c_attn.weight'] = torch.cat([gw[:, 2 * d:], gw[:, :d], gw[:, d:2 * d]], 1)        # (Q, K, V) -> (V, Q, K)
        g[p_ + 'attn.c_attn.bias'] = torch.cat([gb[2 * d:], gb[:d], gb[d:2 * d]])
        for a, mod in (('ln_1', blk.ln_1), ('ln_2', blk.ln_2)):
            g[p_ + a + '.gamma'], g[p_ + a + '.beta'] = mod.weight.grad, mod.bias.grad
        for a, mod in (('attn.c_proj', blk.attn.c_proj), ('mlp.c_fc', blk.mlp.c_fc), ('mlp.c_proj', blk.mlp.c_proj)):
            g[p_ + a + '.weight'], g[p_ + a + '.bias'] = mod.weight.grad, mod.bias.grad
    for n, c in mods.items():
        g[n + '.weight'], g[n + '.bias'] = c.weight.grad, c.bias.grad
    return dict(loss=float(loss), ce=float(ce.mean()), pos=float(pos.mean()), ori=float(ori.mean())), g


def main():
    res = {}
    # ---- tiny: the HIP kernels' small test shape (tests/conftest.py TINY_MIGT + a localization head)
    tiny = dict(n_embeddings=64, n_head=2, d_model=128, n_layer=2, token_image_size=4, sequence_size=4, localization_weight='1', pose_multiplier=0.2)
    cfg = MIGTConfig(**tiny)
    sd = make_migt_weights(cfg, seed=11, std=0.08)
    g = np.random.default_rng(5)
    ids = torch.from_numpy(g.integers(0, 64, size=(2, 4, 4, 4)))
    _, cams_raw = synthetic_scene_batch(2, 4, 8, 9)
    cams = relative_normalized(cams_raw)
    res.update(tiny_seed=11, tiny_std=0.08, tiny_ids=ids.numpy(), tiny_cams_raw=cams_raw.astype(np.float32), tiny_cams=cams.numpy().astype(np.float32))
    res['tiny_logits'] = run(cfg, sd, ids, cams, False)['logits'].numpy().astype(np.float32)
    mo = run(cfg, sd, ids, cams, True)
    res['tiny_multi_logits'] = mo['logits'].numpy().astype(np.float32)
    res['tiny_multi_pose'] = mo['pose_prediction'].numpy().astype(np.float32)
    # ---- the training graph with autograd (tiny shape, constant localization weight): loss terms, and per variable the gradient's L2 norm and 48
    # entries at fixed pseudo-random positions (the whole gradient would be 3.4 MB)
    tcfg = MIGTConfig(**dict(tiny, localization_weight='2', n_loss_skip=1))
    lossd, grads = train_graph(tcfg, sd, ids, cams)
    rng = np.random.default_rng(77)
    res.update(train_loss=lossd['loss'], train_ce=lossd['ce'], train_pos=lossd['pos'], train_ori=lossd['ori'])
    names = sorted(grads)
    res['train_names'] = np.array(names)
    res['train_norms'] = np.array([float(grads[n].norm()) for n in names])
    idx = [rng.integers(0, grads[n].numel(), size=48) for n in names]
    res['train_idx'] = np.stack(idx)
    res['train_samples'] = np.stack([grads[n].reshape(-1)[torch.from_numpy(i)].numpy() for n, i in zip(names, idx)])
    # ---- full size: the bench's transformer (SM7: 6 context views + the MASK view), last-view logits
    cfg = MIGTConfig(sequence_size=6, n_loss_skip=1, pose_multiplier=0.2, localization_weight='cosine(0,1,120000)')
    sd = make_migt_weights(cfg, seed=0)
    ids = torch.from_numpy(g.integers(0, 1024, size=(1, 7, 8, 8)))
    ids[:, -1] = 1024                                          # MASK view
    _, cams_raw = synthetic_scene_batch(1, 7, 8, 21)
    cams = relative_normalized(cams_raw)
    full = run(cfg, sd, ids, cams, False)['logits'][:, -1]
    res.update(full_seed=0, full_ids=ids.numpy(), full_cams=cams.numpy().astype(np.float32), full_logits_last=full.numpy().astype(np.float32))
    # ---- full size, the LOCALIZATION pass of the evaluator (evaluate_transformer.py:134-140; migt.py:369,387-390,430-451): real codes in every view,
    # cameras of the 6 context views only — the last view carries the LOC token's embedding instead of a pose embedding — and the camera head on it
    codes = torch.from_numpy(g.integers(0, 1024, size=(1, 7, 8, 8)))
    m = build_gpt2(cfg, sd)
    wte = t(sd, 'wte.weight')
    pin = torch.cat([cams[:, :-1, :3] * cfg.pose_multiplier, cams[:, :-1, 3:]], -1)
    pe = torch.cat([hf_mlp(sd, 'pose_embedding', pin), wte[cfg.n_embeddings + 1].reshape(1, 1, -1)], 1)[:, :, None, :]        # [1,7,1,d]
    emb = wte[codes.reshape(1, 7, 64)] + pe
    view = torch.arange(7).repeat_interleave(64)
    mask = torch.where(view[:, None] >= view[None, :], 0.0, -1e4).double()[None, None]
    with torch.no_grad():
        h = m(inputs_embeds=emb.reshape(1, 448, -1), position_ids=torch.arange(64).repeat(7)[None], attention_mask=mask).last_hidden_state.reshape(1, 7, 64, -1)
    y = hf_mlp(sd, 'pose_criterion.pose_classifier', h[:, -1])
    qn = y[..., 3:] * torch.rsqrt((y[..., 3:] ** 2).sum(-1, keepdim=True).clamp_min(1e-12))
    qn = qn * (2 * (qn[..., :1] >= 0).double() - 1)
    res.update(full_loc_codes=codes.numpy(), full_loc_pose_last=torch.cat([y[..., :3] / cfg.pose_multiplier, qn], -1).numpy().astype(np.float32))
    out = os.path.join(REPO, 'tests', 'golden', 'migt_hf_gpt2.npz')
    np.savez_compressed(out, **res)
    print('wrote', out, {k: getattr(v, 'shape', v) for k, v in res.items()})


if __name__ == '__main__':
    main()
