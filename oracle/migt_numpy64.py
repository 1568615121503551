"""ORACLE (test infrastructure, NOT product code) — second, INDEPENDENT restatement of the reference transformer, fp64 numpy.

Purpose (VERDICT r1 "weak" #1): ``oracle/migt_oracle.py`` is the restatement every GPU parity test of the
transformer compares with, and TensorFlow — the only thing that could pin it — is absent from the image (PARITY UNPINNED, see that
file's header).  This file is a second opinion: written from the reference source alone, op for op in the order TensorFlow
executes them, in numpy / float64, sharing NO code with ``migt_oracle.py`` (no imports from it, different data structures: plain
numpy arrays, per-stream Python lists like the reference's ``hidden_states`` list, the attention in the reference's own
w_old / w_new / concat / split algebra).  ``tests/test_oracle_migt.py`` diffs the two on inference, multi-stream and training-loss
graphs.  Two restatements by one author agreeing is weaker evidence than a TensorFlow run — it removes transcription slips, not a
shared misreading — so the MIGT half stays "parity unpinned".

Only ``tests/`` may import this module.

Reference lines followed (paths relative to /root/reference/viewformer):
  models/migt.py:17-56 SharedEmbeddings, :59-73 MLP, :76-96 Conv1D, :99-104 label-smoothed CE, :123-129 quaternion_reduce_mean,
  :132-179 QuaternionPoseRepresentation, :182-217 BranchingAttention, :220-238 Block, :338-455 MIGT.call
  models/branching_attention.py:5-18 compute_attention, :41-61 compute_causal_block_attention,
  :82-126 compute_causal_block_multiend_attention
  utils/geometry_tf.py:44-50 quaternion_normalize / quaternion_remove_sign
Third-party semantics restated from TensorFlow 2.4.1's documented behaviour (requirements.txt:4; source not in the tree):
tf.nn.gelu(approximate=False) = 0.5 x (1 + erf(x / sqrt 2)); LayerNormalization = (x - mean) * rsqrt(var + eps) * gamma + beta with
the biased variance over the last axis; tf.nn.softmax = exp(x - max) / sum; tf.linalg.l2_normalize = x * rsqrt(max(sum x^2, eps));
tf.losses.mse = mean over the last axis of the squared difference.
"""
import math

import numpy as np
from scipy.special import erf as _erf

F = np.float64
EPS_LN = 1e-5                                   # migt.py:14


def _w(sd, name):
    v = sd[name]
    v = v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)
    return v.astype(F)


# ---------------------------------------------------------------- layers (migt.py:17-96)
def tf_gelu(x):
    return 0.5 * x * (1.0 + _erf(x / math.sqrt(2.0)))


def tf_softmax(x):
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def keras_layer_norm(x, gamma, beta):
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    return (x - mean) / np.sqrt(var + EPS_LN) * gamma + beta


def conv1d(sd, prefix, x):
    """Conv1D.call, migt.py:89-96: reshape to [-1, nx], x @ weight[nx, nf] + bias[1, nf], reshape back"""
    w, b = _w(sd, prefix + '.weight'), _w(sd, prefix + '.bias').reshape(1, -1)
    lead = x.shape[:-1]
    y = x.reshape(-1, w.shape[0]) @ w + b
    return y.reshape(*lead, w.shape[1])


def mlp_call(sd, prefix, x):
    """MLP.call, migt.py:69-73 (dropout is the identity with training=False)"""
    return conv1d(sd, prefix + '.c_proj', tf_gelu(conv1d(sd, prefix + '.c_fc', x)))


# ---------------------------------------------------------------- attention (branching_attention.py)
def compute_attention(k, v, q, attention_mask=None):
    """branching_attention.py:5-18"""
    w = q @ np.swapaxes(k, -1, -2)                                   # tf.matmul(q, k, transpose_b=True): NO 1/sqrt(d)
    if attention_mask is not None:
        nd, ns = w.shape[-2:]
        m = attention_mask.reshape((1,) * (w.ndim - 2) + (nd, ns))
        w = w * m - 1e4 * (1 - m)
    return tf_softmax(w) @ v


def compute_causal_block_attention(k, v, q):
    """branching_attention.py:41-61"""
    b, h, ns, l, dh = k.shape
    nd = q.shape[-3]
    i = np.repeat(np.arange(nd), l)[:, None]
    j = np.repeat(np.arange(ns), l)
    m = (i >= j - ns + nd).astype(F)
    a = compute_attention(k.reshape(b, h, ns * l, dh), v.reshape(b, h, ns * l, dh), q.reshape(b, h, nd * l, dh), m)
    return a.reshape(b, h, nd, l, dh)


def compute_causal_block_multiend_attention(kset, vset, qset):
    """branching_attention.py:82-126"""
    k, v = kset[0], vset[0]
    outputs = [compute_causal_block_attention(k, v, qset[0])]
    b, h, ns, l, dh = k.shape
    k_flat = k[:, :, :-1].reshape(b, h, (ns - 1) * l, dh)
    v_flat = v[:, :, :-1].reshape(b, h, (ns - 1) * l, dh)
    nd = qset[0].shape[-3]
    i = np.repeat(np.arange(nd), l)[:, None]
    j = np.repeat(np.arange(ns - 1), l)
    m = (i >= j - ns + nd + 1).astype(F).reshape(1, 1, nd * l, (ns - 1) * l)
    for k_new, v_new, q in zip(kset[1:], vset[1:], qset[1:]):
        nd = q.shape[-3]
        q_flat = q.reshape(b, h, nd * l, dh)
        w_old = q_flat @ np.swapaxes(k_flat, -1, -2)
        w_old = w_old * m - 1e4 * (1 - m)
        w_new = (q @ np.swapaxes(k_new, -1, -2)).reshape(b, h, -1, l)       # per view: its own l x l block
        w = tf_softmax(np.concatenate([w_old, w_new], -1))
        attn_old = (w[:, :, :, :(ns - 1) * l] @ v_flat).reshape(b, h, nd, l, dh)
        w_new = w[:, :, :, (ns - 1) * l:].reshape(b, h, nd, l, l)
        attn_new = np.einsum('ijklm,ijkmv->ijklv', w_new, v_new)
        outputs.append(attn_old + attn_new)
    return outputs


def _split_heads(x, n_head):
    """BranchingAttention.split_heads, migt.py:201-205: [B,S,L,d] -> [B,H,S,L,d/H]"""
    B, S, L, d = x.shape
    return x.reshape(B, S, L, n_head, d // n_head).transpose(0, 3, 1, 2, 4)


def _merge_heads(x):
    """BranchingAttention.merge_heads, migt.py:195-199"""
    x = x.transpose(0, 2, 3, 1, 4)
    return x.reshape(*x.shape[:-2], x.shape[-2] * x.shape[-1])


def branching_attention_call(sd, prefix, xs, n_head):
    """BranchingAttention.call, migt.py:207-217: tf.split(c_attn(x), 3) is unpacked as (v, q, k) — in that order"""
    vs, qs, ks = [], [], []
    for x in xs:
        third = np.split(conv1d(sd, prefix + '.c_attn', x), 3, axis=-1)
        vs.append(_split_heads(third[0], n_head))
        qs.append(_split_heads(third[1], n_head))
        ks.append(_split_heads(third[2], n_head))
    a = compute_causal_block_multiend_attention(ks, vs, qs)
    return [conv1d(sd, prefix + '.c_proj', _merge_heads(t)) for t in a]


def block_call(sd, prefix, xs, n_head):
    """Block.call, migt.py:230-238"""
    a = [keras_layer_norm(x, _w(sd, prefix + '.ln_1.gamma'), _w(sd, prefix + '.ln_1.beta')) for x in xs]
    a = branching_attention_call(sd, prefix + '.attn', a, n_head)
    xs = [x + y for x, y in zip(xs, a)]
    m = [keras_layer_norm(x, _w(sd, prefix + '.ln_2.gamma'), _w(sd, prefix + '.ln_2.beta')) for x in xs]
    m = [mlp_call(sd, prefix + '.mlp', t) for t in m]
    return [x + y for x, y in zip(xs, m)]


# ---------------------------------------------------------------- pose representation (migt.py:123-179)
def quaternion_normalize(x, epsilon=1e-12):
    return x / np.sqrt(np.maximum((x * x).sum(-1, keepdims=True), epsilon))


def quaternion_remove_sign(x):
    return x * (2.0 * (x[..., :1] >= 0).astype(F) - 1.0)


def get_model_input(poses, position_multiplier, pose_multiplier):
    """QuaternionPoseRepresentation.get_model_input, migt.py:139-145"""
    assert poses.shape[-1] == 7
    xyz, quat = poses[..., :3], poses[..., 3:]
    xyz = xyz * position_multiplier
    xyz = xyz * pose_multiplier.reshape((-1,) + (1,) * (xyz.ndim - 1))
    return np.concatenate([xyz, quat], -1)


def pose_criterion_call(sd, internal_x, position_multiplier, pose_multiplier, y=None, skip_first=None):
    """QuaternionPoseRepresentation.call, migt.py:156-179"""
    internal_x = mlp_call(sd, 'pose_criterion.pose_classifier', internal_x)
    xyz, quaternion = internal_x[..., :3], internal_x[..., 3:]
    xyz = xyz / pose_multiplier.reshape((-1,) + (1,) * (xyz.ndim - 1))
    qn = quaternion_remove_sign(quaternion_normalize(quaternion))
    output_x = np.concatenate([xyz / position_multiplier, qn], -1)
    if y is None:
        return output_x
    y = y * np.array([position_multiplier] * 3 + [1] * 4, dtype=F)
    position_loss = ((y[..., :3] - xyz) ** 2).mean(-1)                       # tf.losses.mse
    orientation_loss = ((y[..., 3:] - quaternion) ** 2).mean(-1)             # against the RAW quaternion, :171
    if skip_first is not None:
        position_loss = position_loss[:, skip_first:]
        orientation_loss = orientation_loss[:, skip_first:]
    return output_x, position_loss.mean(axis=(1, 2)), orientation_loss.mean(axis=(1, 2))


def reduce_cameras(x, axis=-2):
    """QuaternionPoseRepresentation.reduce + quaternion_reduce_mean, migt.py:123-129,150-154"""
    xyz, quat = x[..., :3], x[..., 3:]
    xyz = xyz.mean(axis)
    quat = quaternion_remove_sign(quaternion_normalize(quat)).mean(axis)
    quat = quaternion_remove_sign(quaternion_normalize(quat))
    return np.concatenate([xyz, quat], -1)


# ---------------------------------------------------------------- MIGT.call (migt.py:338-455)
def migt_call(sd, cfg, inputs, compute_losses=False, train_counter=0, random_pose_multiplier=None):
    """training=False graph (dropout off); ``random_pose_multiplier`` [B] defaults to ones (:352-353).  Returns numpy fp64."""
    n_embeddings, n_head = cfg.n_embeddings, cfg.n_head
    mask_token, localization_token = n_embeddings, n_embeddings + 1          # :256-257
    from viewformer_amd.schedules import parse                                # host-side scalar schedule (utils/schedules.py)
    loc_sched = parse(cfg.localization_weight).with_total_steps(cfg.total_steps)
    use_localization = not loc_sched.is_zero()                               # :268-269
    wte, wpe = _w(sd, 'wte.weight'), _w(sd, 'wpe.embeddings')

    poses = np.asarray(inputs['poses'])
    assert poses.dtype == np.float32, 'tf.debugging.assert_type(poses, tf.float32), :346'
    poses = poses.astype(F)
    input_ids = np.asarray(inputs['input_ids'])
    original_input_shape = list(input_ids.shape)
    input_ids = input_ids.reshape(input_ids.shape[0], input_ids.shape[1], -1).astype(np.int64)
    input_shape = list(input_ids.shape)
    localization_tokens = inputs.get('localization_tokens')
    output_poses = inputs.get('output_poses')
    B = poses.shape[0]
    rpm = np.ones((B,), F) if random_pose_multiplier is None else np.asarray(random_pose_multiplier, F)

    def embed_pose(p):
        e = mlp_call(sd, 'pose_embedding', get_model_input(p, cfg.pose_multiplier, rpm))      # :354
        return e[..., None, :]                                                                  # expand_dims(-2), :355

    pose_embeddings = embed_pose(poses)
    position_ids = np.arange(0, cfg.token_image_size ** 2)[None, None, :]
    position_embeds = wpe[position_ids]                                      # :358-359
    inputs_embeds = wte[input_ids]                                           # :361
    localization_embeds = output_pose_embeddings = None
    gen_images_pointer = gen_poses_pointer = 0
    loc_seq_size = inputs_embeds.shape[1] - pose_embeddings.shape[1]        # :369

    if compute_losses:                                                       # :371-377
        if localization_tokens is None and use_localization:
            localization_tokens, localization_embeds = input_ids, inputs_embeds
        if output_poses is None:
            output_poses, output_pose_embeddings = poses, pose_embeddings
    if localization_tokens is not None and localization_embeds is None:      # :378-381
        lt = np.asarray(localization_tokens)
        lt = lt.reshape(lt.shape[0], lt.shape[1], -1).astype(np.int64)
        localization_embeds = wte[lt]
    if output_poses is not None and output_pose_embeddings is None:          # :382-385
        output_pose_embeddings = embed_pose(np.asarray(output_poses).astype(F))

    if use_localization and not compute_losses:                              # :387-390
        lpe = np.broadcast_to(wte[localization_token].reshape(1, 1, 1, -1),
                              (inputs_embeds.shape[0], loc_seq_size, pose_embeddings.shape[-2], wte.shape[1]))
        pose_embeddings = np.concatenate([pose_embeddings, lpe], 1)

    hidden_states = [inputs_embeds + position_embeds + pose_embeddings]      # :392
    if output_pose_embeddings is not None:                                   # :393-396
        mask_embeds = wte[mask_token].reshape(1, 1, 1, -1)
        hidden_states.append(mask_embeds + position_embeds + output_pose_embeddings)
        gen_images_pointer = len(hidden_states) - 1
    if localization_embeds is not None:                                      # :398-401
        loc_tok = wte[localization_token].reshape(1, 1, 1, -1)
        hidden_states.append(localization_embeds + position_embeds + loc_tok)
        gen_poses_pointer = len(hidden_states) - 1

    output_shape = input_shape + [hidden_states[0].shape[-1]]
    for i in range(cfg.n_layer):                                             # :405-406
        hidden_states = block_call(sd, f'h.{i}', hidden_states, n_head)
    g, bta = _w(sd, 'ln_f.gamma'), _w(sd, 'ln_f.beta')
    hidden_states = [keras_layer_norm(x, g, bta).reshape(output_shape) for x in hidden_states]   # :408-409

    out = dict(hidden_states=hidden_states)
    loss = 0
    hx = hidden_states[gen_images_pointer]
    lm_logits = (hx.reshape(-1, wte.shape[1]) @ wte.T).reshape(*hx.shape[:-1], wte.shape[0])[..., :n_embeddings]   # :417, :51-56
    if compute_losses:                                                       # :418-428
        z = lm_logits - lm_logits.max(-1, keepdims=True)
        logp = z - np.log(np.exp(z).sum(-1, keepdims=True))
        onehot = np.eye(n_embeddings, dtype=F)[input_ids]
        if cfg.label_smoothing > 0:                                          # :99-104
            onehot = onehot * (1.0 - cfg.label_smoothing) + cfg.label_smoothing / n_embeddings
        ce_loss = -(onehot * logp).sum(-1)
        ce_loss = ce_loss[:, cfg.n_loss_skip:].mean(axis=(1, 2))
        out['ce_loss'] = ce_loss
        loss = loss + ce_loss * cfg.image_generation_weight

    if use_localization:                                                     # :430-451
        poses_input = hidden_states[gen_poses_pointer]
        if compute_losses:
            gt_poses = poses[..., None, :]
            poses_out, pos_l, ori_l = pose_criterion_call(sd, poses_input, cfg.pose_multiplier, rpm, gt_poses, cfg.n_loss_skip)
            if cfg.use_dynamic_pose_loss:                                    # DynamicLossWeightingCriterion.call, :114-118
                s = _w(sd, 'pose_loss_weighting_criterion.pos_ori_weights')
                pose_loss = (s + np.exp(-s) * np.stack([pos_l, ori_l], -1)).sum()
            else:
                pose_loss = pos_l + ori_l
            lw = float(loc_sched(train_counter))
            loss = loss + pose_loss * lw
            out.update(pose_loss=pose_loss, pose_pos_loss=pos_l, pose_ori_loss=ori_l, localization_weight=lw)
        else:
            poses_out = pose_criterion_call(sd, poses_input, cfg.pose_multiplier, rpm)
        out['pose_prediction'] = poses_out
    out['logits'] = lm_logits.reshape(original_input_shape + [-1])
    out['loss'] = loss
    return out


# ---------------------------------------------------------------- evaluator camera frames (evaluate_transformer.py:70-94)
def quaternion_multiply(q1, q2):
    """geometry_tf.py:6-13, (w, x, y, z)"""
    w1, x1, y1, z1 = np.moveaxis(q1, -1, 0)
    w2, x2, y2, z2 = np.moveaxis(q2, -1, 0)
    return np.stack((-x1 * x2 - y1 * y2 - z1 * z2 + w1 * w2,
                     x1 * w2 + y1 * z2 - z1 * y2 + w1 * x2,
                     -x1 * z2 + y1 * w2 + z1 * x2 + w1 * y2,
                     x1 * y2 - y1 * x2 + z1 * w2 + w1 * z2), -1)


def _conj(q):
    return np.concatenate([q[..., :1], -q[..., 1:]], -1)


def _rotate(point, q):
    p = np.concatenate([np.zeros_like(point[..., :1]), point], -1)
    return quaternion_multiply(quaternion_multiply(q, p), _conj(q))[..., 1:]


def to_relative_cameras(cameras):
    """evaluate_transformer.py:70-78"""
    xyz, quat = cameras[..., :3], cameras[..., 3:]
    t_xyz, t_quat = xyz[..., :1, :], quat[..., :1, :]
    rinv = np.broadcast_to(_conj(t_quat), quat.shape)
    return (np.concatenate([_rotate(xyz - t_xyz, rinv), quaternion_multiply(rinv, quat)], -1),
            np.concatenate([t_xyz, t_quat], -1))


def normalize_cameras(cameras):
    """evaluate_transformer.py:90-94"""
    return np.concatenate([cameras[..., :3], quaternion_remove_sign(quaternion_normalize(cameras[..., 3:]))], -1)
