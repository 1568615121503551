"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the
reference's MIGT image-token transformer and of the evaluator's pose pre/post-
processing.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.

PARITY UNPINNED: the reference transformer is TensorFlow/Keras
(viewformer/models/migt.py, branching_attention.py; tensorflow==2.4.1 per
requirements.txt:4) and TensorFlow is absent from this image, so this file
cannot be checked against the reference's own outputs.  It follows the
reference line by line (citations below) and is pinned only by the structural
invariants of SURVEY.md §8(c) (tests/test_oracle_migt.py): single-stream ==
dense masked attention, branch equivalence, train/infer consistency, fp64 arm.
Since round 5 it is additionally ANCHORED TO A THIRD-PARTY IMPLEMENTATION (still
not the reference): tests/golden/migt_hf_gpt2.npz holds outputs of Hugging Face
``transformers``' GPT-2 on MIGT weights — no score scaling, (V, Q, K) split,
block-causal 0 / -1e4 mask expressed through GPT-2's own config and inputs
(tests/golden/make_hf_gpt2_golden.py) — which this file reproduces to 2e-7
(single-stream and multi-stream logits, camera predictions, full-size logits)
and, through train_oracle.py, to 1e-8 of every gradient of the training graph.
Third-party arithmetic restated from its documented behaviour: tf.nn.gelu
(exact erf form), LayerNormalization(eps=1e-5), tf.nn.softmax,
tf.linalg.l2_normalize(eps=1e-12): x * rsqrt(max(sum(x^2), eps)).
"""
import math
import torch
import torch.nn.functional as F


def _t(sd, name, dtype):
    v = sd[name]
    if not isinstance(v, torch.Tensor):
        v = torch.from_numpy(v)
    return v.to(dtype)


# ------------------------------------------------------------------ geometry (geometry_tf.py)
def quaternion_multiply(q1, q2):
    """geometry_tf.py:6-13 (w, x, y, z order)."""
    w1, x1, y1, z1 = q1.unbind(-1)
    w2, x2, y2, z2 = q2.unbind(-1)
    x = x1 * w2 + y1 * z2 - z1 * y2 + w1 * x2
    y = -x1 * z2 + y1 * w2 + z1 * x2 + w1 * y2
    z = x1 * y2 - y1 * x2 + z1 * w2 + w1 * z2
    w = -x1 * x2 - y1 * y2 - z1 * z2 + w1 * w2
    return torch.stack((w, x, y, z), -1)


def quaternion_normalize(x, epsilon=1e-12):
    """geometry_tf.py:44-45 — tf.linalg.l2_normalize."""
    sq = (x * x).sum(-1, keepdim=True)
    return x * torch.rsqrt(torch.clamp(sq, min=epsilon))


def quaternion_remove_sign(x):
    """geometry_tf.py:48-50"""
    sign = 2 * (x[..., :1] >= 0).to(x.dtype) - 1
    return x * sign


def quaternion_conjugate(q):
    """geometry_tf.py:53-68"""
    return torch.cat((q[..., :1], -q[..., 1:]), -1)


def quaternion_rotate(point, q):
    """geometry_tf.py:71-91"""
    point = torch.cat([torch.zeros_like(point[..., :1]), point], -1)
    point = quaternion_multiply(q, point)
    point = quaternion_multiply(point, quaternion_conjugate(q))
    return point[..., 1:]


def to_relative_cameras(cameras):
    """evaluate_transformer.py:70-78"""
    xyz, quat = cameras[..., :3], cameras[..., 3:]
    t_xyz, t_quat = xyz[..., :1, :], quat[..., :1, :]
    rinv = quaternion_conjugate(t_quat)
    xyz = quaternion_rotate(xyz - t_xyz, rinv.expand_as(quat))
    quat = quaternion_multiply(rinv.expand_as(quat), quat)
    return torch.cat((xyz, quat), -1), torch.cat((t_xyz, t_quat), -1)


def from_relative_cameras(cameras, transform):
    """evaluate_transformer.py:81-87"""
    t_xyz, t_quat = transform[..., :3], transform[..., 3:]
    xyz, quat = cameras[..., :3], cameras[..., 3:]
    quat2 = quaternion_multiply(t_quat.expand_as(quat), quat)
    xyz = quaternion_rotate(xyz, t_quat.expand_as(quat)) + t_xyz
    return torch.cat((xyz, quat2), -1)


def normalize_cameras(cameras):
    """evaluate_transformer.py:90-94"""
    xyz, quat = cameras[..., :3], cameras[..., 3:]
    return torch.cat((xyz, quaternion_remove_sign(quaternion_normalize(quat))), -1)


def quaternion_reduce_mean(quat, axis=-2):
    """migt.py:123-129"""
    quat = quaternion_remove_sign(quaternion_normalize(quat))
    quat = quat.mean(axis)
    return quaternion_remove_sign(quaternion_normalize(quat))


def reduce_cameras(x, axis=-2):
    """QuaternionPoseRepresentation.reduce, migt.py:150-154 (== MIGT.reduce_cameras :532)."""
    return torch.cat((x[..., :3].mean(axis), quaternion_reduce_mean(x[..., 3:], axis)), -1)


# ------------------------------------------------------------------ layers
# Training-mode dropout hook (tests only): ``with dropout_masks(obj):`` makes the forward multiply by obj's masks at the four
# Dropout sites of the reference (migt.py:72 MLP, :216 resid, :403 embeddings; branching_attention.py:15-17 attention weights).
# obj.elem(kind, layer, stream, x) -> mask like x;  obj.attn(layer, stream, w, key_index) -> mask like w (see train_oracle).
_DROP = None


class dropout_masks:
    def __init__(self, obj):
        self.obj = obj

    def __enter__(self):
        global _DROP
        self.prev, _DROP = _DROP, self.obj
        return self.obj

    def __exit__(self, *exc):
        global _DROP
        _DROP = self.prev


def gelu(x):
    """tf.nn.gelu default (approximate=False): 0.5 x (1 + erf(x / sqrt 2))."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def conv1d(sd, name, x, dtype):
    """Conv1D.call, migt.py:89-96: x @ W[nx, nf] + b."""
    return x @ _t(sd, name + '.weight', dtype) + _t(sd, name + '.bias', dtype)


def mlp(sd, name, x, dtype):
    """MLP.call, migt.py:69-73 (dropout = identity at inference)."""
    return conv1d(sd, name + '.c_proj', gelu(conv1d(sd, name + '.c_fc', x, dtype)), dtype)


def layer_norm(sd, name, x, dtype):
    """tf.keras LayerNormalization(epsilon=1e-5), migt.py:14,225,227,292."""
    return F.layer_norm(x, (x.shape[-1],), _t(sd, name + '.gamma', dtype), _t(sd, name + '.beta', dtype), eps=1e-5)


# ------------------------------------------------------------------ attention (branching_attention.py)
def compute_attention(k, v, q, attention_mask=None, wmask=None):
    """branching_attention.py:5-18 — no 1/sqrt(d) scale; mask as w*m - 1e4*(1-m); ``wmask`` = attn_dropout (:15-17)."""
    w = q @ k.transpose(-1, -2)
    if attention_mask is not None:
        w = w * attention_mask - 1e4 * (1 - attention_mask)
    w = torch.softmax(w, dim=-1)
    if wmask is not None:
        w = w * wmask
    return w @ v


def compute_causal_block_attention(k, v, q, wmask=None):
    """branching_attention.py:41-61 — block-causal over views, full inside a view."""
    b, h, ns, l, _ = k.shape
    nd = q.shape[-3]
    i = torch.arange(nd).repeat_interleave(l)[:, None]
    j = torch.arange(ns).repeat_interleave(l)
    m = (i >= j - ns + nd).to(k.dtype)
    a = compute_attention(k.reshape(b, h, ns * l, -1), v.reshape(b, h, ns * l, -1),
                          q.reshape(b, h, nd * l, -1), attention_mask=m, wmask=wmask)
    return a.reshape(b, h, nd, l, -1)


def compute_causal_block_multiend_attention(kset, vset, qset, layer=None):
    """branching_attention.py:82-126.  With a dropout hook installed, ``layer`` selects the attention-dropout masks."""
    k, v = kset[0], vset[0]
    drop = _DROP if layer is not None else None
    outputs = [compute_causal_block_attention(k, v, qset[0], wmask=drop.attn(layer, 0, 'main') if drop else None)]
    b, h, ns, l, dh = k.shape
    # (explicit head dim instead of the reference's -1 so that a one-view sequence, which
    # the reference cannot reshape, is still defined here)
    k_flat = k[:, :, :-1].reshape(b, h, (ns - 1) * l, dh)
    v_flat = v[:, :, :-1].reshape(b, h, (ns - 1) * l, dh)
    nd = qset[0].shape[-3]
    i = torch.arange(nd).repeat_interleave(l)[:, None]
    j = torch.arange(ns - 1).repeat_interleave(l)
    m = (i >= j - ns + nd + 1).to(k.dtype).reshape(1, 1, nd * l, (ns - 1) * l)
    for k_new, v_new, q in zip(kset[1:], vset[1:], qset[1:]):
        nd = q.shape[-3]
        q_flat = q.reshape(b, h, nd * l, -1)
        w_old = q_flat @ k_flat.transpose(-1, -2)
        w_old = w_old * m - 1e4 * (1 - m)
        w_new = (q @ k_new.transpose(-1, -2)).reshape(b, h, -1, l)
        w = torch.softmax(torch.cat([w_old, w_new], -1), dim=-1)
        if drop:
            w = w * drop.attn(layer, len(outputs), 'branch')
        attn_old = (w[:, :, :, :(ns - 1) * l] @ v_flat).reshape(b, h, nd, l, -1)
        w_new = w[:, :, :, (ns - 1) * l:].reshape(b, h, nd, l, l)
        attn_new = torch.einsum('ijklm,ijkmv->ijklv', w_new, v_new)
        outputs.append(attn_old + attn_new)
    return outputs


def _split_heads(x, n_head):
    """migt.py:201-205: [B,S,L,D] -> [B,H,S,L,D/H]."""
    b, s, l, d = x.shape
    return x.reshape(b, s, l, n_head, d // n_head).permute(0, 3, 1, 2, 4)


def _merge_heads(x):
    """migt.py:195-199"""
    b, h, s, l, dh = x.shape
    return x.permute(0, 2, 3, 1, 4).reshape(b, s, l, h * dh)


def _layer_of(name):
    return int(name.split('.')[1]) if _DROP is not None else None


def branching_attention(sd, name, xs, n_head, dtype):
    """BranchingAttention.call, migt.py:207-217: c_attn thirds are (V, Q, K)."""
    vs, qs, ks = [], [], []
    for x in xs:
        y = conv1d(sd, name + '.c_attn', x, dtype)
        v, q, k = y.chunk(3, dim=-1)
        vs.append(_split_heads(v, n_head))
        qs.append(_split_heads(q, n_head))
        ks.append(_split_heads(k, n_head))
    a = compute_causal_block_multiend_attention(ks, vs, qs, layer=_layer_of(name))
    out = [conv1d(sd, name + '.c_proj', _merge_heads(y), dtype) for y in a]
    if _DROP is not None:                                       # resid_dropout, migt.py:216
        out = [y * _DROP.elem('resid', _layer_of(name), s, y) for s, y in enumerate(out)]
    return out


def block(sd, name, xs, n_head, dtype):
    """Block.call, migt.py:230-238."""
    a = branching_attention(sd, name + '.attn', [layer_norm(sd, name + '.ln_1', x, dtype) for x in xs], n_head, dtype)
    xs = [x + y for x, y in zip(xs, a)]
    m = [mlp(sd, name + '.mlp', layer_norm(sd, name + '.ln_2', x, dtype), dtype) for x in xs]
    if _DROP is not None:                                       # MLP dropout, migt.py:72
        m = [y * _DROP.elem('mlp', _layer_of(name), s, y) for s, y in enumerate(m)]
    return [x + y for x, y in zip(xs, m)]


def pose_model_input(poses, position_multiplier):
    """QuaternionPoseRepresentation.get_model_input, migt.py:139-145 (inference: multiplier 1)."""
    return torch.cat([poses[..., :3] * position_multiplier, poses[..., 3:]], -1)


def pose_head(sd, cfg, hidden, dtype):
    """QuaternionPoseRepresentation.call without targets, migt.py:156-164,178-179."""
    y = mlp(sd, 'pose_criterion.pose_classifier', hidden, dtype)
    xyz, quat = y[..., :3], y[..., 3:]
    xyz = xyz / 1.0          # random_pose_multiplier == 1 at inference (migt.py:353,160-161)
    quat = quaternion_remove_sign(quaternion_normalize(quat))
    return torch.cat([xyz / cfg.pose_multiplier, quat], -1)


def migt_forward(sd, cfg, input_ids, poses, localization_tokens=None, output_poses=None,
                 dtype=torch.float32, compute_losses=False, grad=False, pose_factors=None):
    """MIGT.call (training=False), migt.py:338-455.  ``pose_factors`` [B]: the training-mode random pose multiplier (:350-354; ones
    at inference) applied to the model-input positions (:141-144).

    input_ids [B,S,t,t] int, poses [B,Sp,7] float32.  Returns dict with
    ``logits`` [B,S,t,t,n_embeddings], ``hidden_states`` (list per stream,
    [B,S,L,d]) and, when the model uses localization, ``pose_prediction``
    [B,S,L,7].  ``compute_losses=True`` builds the 2-/3-stream training graph
    (loss values themselves are not restated here).
    """
    with (torch.enable_grad() if grad else torch.no_grad()):     # grad=True: autograd reference for the training step
        B, S = input_ids.shape[:2]
        ids = input_ids.reshape(B, S, -1).long()
        L = ids.shape[-1]
        wte = _t(sd, 'wte.weight', dtype)
        wpe = _t(sd, 'wpe.embeddings', dtype)
        mask_token, loc_token = cfg.n_embeddings, cfg.n_embeddings + 1
        use_loc = cfg.use_localization
        poses = poses.to(torch.float32)

        def pose_embed(p):
            # pose MLP runs in float32 in the reference (dtype='float32', migt.py:291)
            pin = pose_model_input(p.to(dtype), cfg.pose_multiplier)
            if pose_factors is not None:
                pin = torch.cat([pin[..., :3] * pose_factors.to(dtype).view(-1, 1, 1), pin[..., 3:]], -1)
            e = mlp(sd, 'pose_embedding', pin, dtype)
            return e.unsqueeze(-2)

        pose_emb = pose_embed(poses)                           # [B,Sp,1,d]
        pos_emb = wpe[:L][None, None]                          # :358-359
        tok_emb = wte[ids]                                     # :361
        loc_seq = S - pose_emb.shape[1]                        # :369
        loc_ids, loc_emb, out_pose_emb = localization_tokens, None, None
        if compute_losses:                                     # :371-377
            if loc_ids is None and use_loc:
                loc_ids, loc_emb = ids, tok_emb
            if output_poses is None:
                output_poses, out_pose_emb = poses, pose_emb
        if loc_ids is not None and loc_emb is None:            # :378-381
            loc_emb = wte[loc_ids.reshape(B, loc_ids.shape[1], -1).long()]
        if output_poses is not None and out_pose_emb is None:  # :382-385
            out_pose_emb = pose_embed(output_poses.to(torch.float32))
        if use_loc and not compute_losses:                     # :387-390
            lpe = wte[loc_token].reshape(1, 1, 1, -1).expand(B, loc_seq, 1, -1)
            pose_emb = torch.cat([pose_emb, lpe], 1)

        streams = [tok_emb + pos_emb + pose_emb]               # :392
        img_ptr = pose_ptr = 0
        if out_pose_emb is not None:                           # :393-396
            streams.append(wte[mask_token].reshape(1, 1, 1, -1) + pos_emb + out_pose_emb)
            img_ptr = len(streams) - 1
        if loc_emb is not None:                                # :398-401
            streams.append(loc_emb + pos_emb + wte[loc_token].reshape(1, 1, 1, -1))
            pose_ptr = len(streams) - 1

        if _DROP is not None:                                  # self.drop, :403
            streams = [x * _DROP.elem('embed', 0, s, x) for s, x in enumerate(streams)]
        for i in range(cfg.n_layer):                           # :405-406
            streams = block(sd, f'h.{i}', streams, cfg.n_head, dtype)
        streams = [layer_norm(sd, 'ln_f', x, dtype) for x in streams]   # :408

        out = dict(hidden_states=streams)
        logits = (streams[img_ptr] @ wte.t())[..., :cfg.n_embeddings]   # :417, SharedEmbeddings._linear :51-56
        if use_loc:                                            # :430-451
            out['pose_prediction'] = pose_head(sd, cfg, streams[pose_ptr], dtype)
        out['logits'] = logits.reshape(*input_ids.shape, -1)
        return out
