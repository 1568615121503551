"""Portable deterministic weight generator + parameter-name layout.

There are no trained checkpoints offline (the reference downloads them over
HTTPS, viewformer/utils/_common.py:149-180), so every parity and bench run uses
synthetic weights.  The generator is a counter-based integer PRNG (numpy PCG64
seeded from crc32(tensor name) ^ seed) mapped to fp32, so that this container,
the GPU box, the oracle and the golden-vector script all see bit-identical
tensors without shipping 600 MB of weights.

Parameter names and shapes follow the reference state-dicts so that real
checkpoints map 1:1 later:
  * VQGAN: the Lightning ``state_dict`` keys of viewformer/models/vqgan_th.py
    (``encoder.down.0.block.0.norm1.weight`` ..., conv weights OIHW,
    ``quantize.embeddings`` [D, K]).
  * MIGT: the Keras variable tree of viewformer/models/migt.py:288-315 with '/'
    replaced by '.', (``wte.weight`` [1026, 768], ``wpe.embeddings`` [256, 768],
    ``h.{i}.attn.c_attn.weight`` [768, 2304] stored [nx, nf] as Conv1D does).

Init distributions (reference): conv = PyTorch default (uniform +-1/sqrt(fan_in)),
GroupNorm/LayerNorm gamma=1 beta=0, codebook U(-sqrt3, sqrt3) (utils_th.py:17),
MIGT dense N(0, 0.02) truncated at 2 sigma and zero bias (migt.py:26,85-87,314).
``codebook_scale`` rescales the codebook so that random-init lookups spread over
many codes (with the raw init one or two minimum-norm codes win every row, which
would make index-parity tests vacuous).
"""
import math
import zlib
from collections import OrderedDict

import numpy as np

from .config import VQGANConfig, MIGTConfig


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) << 16) ^ (seed * 0x9E3779B1 & 0xFFFFFFFF)))


def _uniform(name, shape, bound, seed):
    u = _rng(name, seed).random(size=int(np.prod(shape)), dtype=np.float64)
    return ((u * 2.0 - 1.0) * bound).astype(np.float32).reshape(shape)


def _trunc_normal(name, shape, std, seed):
    # Sum of 4 uniforms (Irwin-Hall) -> variance 4/12; rescale to unit variance then
    # clip to +-2 sigma like TF's TruncatedNormal.  Only uniform doubles are drawn,
    # which keeps the stream portable.
    r = _rng(name, seed)
    n = int(np.prod(shape))
    u = r.random(size=(4, n), dtype=np.float64).sum(0) - 2.0
    g = np.clip(u * math.sqrt(3.0), -2.0, 2.0)
    return (g * std).astype(np.float32).reshape(shape)


# ----------------------------------------------------------------------------- VQGAN layout
def _conv(sd, name, cin, cout, k, seed, gain=1.0):
    bound = gain / math.sqrt(cin * k * k)
    sd[name + '.weight'] = _uniform(name + '.weight', (cout, cin, k, k), bound, seed)
    sd[name + '.bias'] = _uniform(name + '.bias', (cout,), bound, seed)


def _norm(sd, name, c, seed, jitter):
    g = np.ones((c,), np.float32)
    b = np.zeros((c,), np.float32)
    if jitter:
        g = g + _uniform(name + '.weight', (c,), 0.25, seed)
        b = b + _uniform(name + '.bias', (c,), 0.25, seed)
    sd[name + '.weight'] = g
    sd[name + '.bias'] = b


def _resblock(sd, name, cin, cout, seed, jitter):
    _norm(sd, name + '.norm1', cin, seed, jitter)
    _conv(sd, name + '.conv1', cin, cout, 3, seed)
    _norm(sd, name + '.norm2', cout, seed, jitter)
    _conv(sd, name + '.conv2', cout, cout, 3, seed)
    if cin != cout:
        _conv(sd, name + '.nin_shortcut', cin, cout, 1, seed)


def _attnblock(sd, name, c, seed, jitter):
    _norm(sd, name + '.norm', c, seed, jitter)
    for p in ('q', 'k', 'v', 'proj_out'):
        _conv(sd, f'{name}.{p}', c, c, 1, seed)


def vqgan_layout(cfg: VQGANConfig):
    """Layer plan shared by weight generation, the oracle and the HIP model.

    Returns ``(encoder_plan, decoder_plan)``: lists of ``(kind, name, args)`` in
    execution order.  kinds: conv3 / down / up / res / attn / norm_swish.
    Mirrors Encoder.__init__/forward (vqgan_th.py:147-225) and
    Decoder.__init__/forward (vqgan_th.py:228-318).
    """
    ch, mult, nrb = cfg.ch, list(cfg.ch_mult), cfg.num_res_blocks
    nres = len(mult)
    enc = [('conv3', 'encoder.conv_in', (cfg.in_channels, ch))]
    res = cfg.image_size
    in_mult = [1] + mult
    bin_ = ch
    for lvl in range(nres):
        bin_ = ch * in_mult[lvl]
        bout = ch * mult[lvl]
        for b in range(nrb):
            enc.append(('res', f'encoder.down.{lvl}.block.{b}', (bin_, bout)))
            bin_ = bout
            if res in cfg.attn_resolutions:
                enc.append(('attn', f'encoder.down.{lvl}.attn.{b}', (bin_,)))
        if lvl != nres - 1:
            enc.append(('down', f'encoder.down.{lvl}.downsample.conv', (bin_, bin_)))
            res //= 2
    enc += [('res', 'encoder.mid.block_1', (bin_, bin_)),
            ('attn', 'encoder.mid.attn_1', (bin_,)),
            ('res', 'encoder.mid.block_2', (bin_, bin_)),
            ('norm_swish', 'encoder.norm_out', (bin_,)),
            ('conv3', 'encoder.conv_out', (bin_, cfg.z_channels))]

    bin_ = ch * mult[nres - 1]
    res = cfg.image_size // 2 ** (nres - 1)
    dec = [('conv3', 'decoder.conv_in', (cfg.z_channels, bin_)),
           ('res', 'decoder.mid.block_1', (bin_, bin_)),
           ('attn', 'decoder.mid.attn_1', (bin_,)),
           ('res', 'decoder.mid.block_2', (bin_, bin_))]
    for lvl in reversed(range(nres)):
        bout = ch * mult[lvl]
        for b in range(nrb + 1):
            dec.append(('res', f'decoder.up.{lvl}.block.{b}', (bin_, bout)))
            bin_ = bout
            if res in cfg.attn_resolutions:
                dec.append(('attn', f'decoder.up.{lvl}.attn.{b}', (bin_,)))
        if lvl != 0:
            dec.append(('up', f'decoder.up.{lvl}.upsample.conv', (bin_, bin_)))
            res *= 2
    dec += [('norm_swish', 'decoder.norm_out', (bin_,)),
            ('conv3', 'decoder.conv_out', (bin_, cfg.out_ch))]
    return enc, dec


def make_vqgan_weights(cfg: VQGANConfig = None, seed: int = 0, codebook_scale: float = None,
                       jitter_norm: bool = True):
    """Deterministic VQGAN state dict (numpy fp32), keys as in the reference."""
    cfg = cfg or VQGANConfig()
    sd = OrderedDict()
    enc, dec = vqgan_layout(cfg)
    for kind, name, args in enc + dec:
        if kind in ('conv3', 'down', 'up'):
            _conv(sd, name, args[0], args[1], 3, seed)
        elif kind == 'res':
            _resblock(sd, name, args[0], args[1], seed, jitter_norm)
        elif kind == 'attn':
            _attnblock(sd, name, args[0], seed, jitter_norm)
        elif kind == 'norm_swish':
            _norm(sd, name, args[0], seed, jitter_norm)
    _conv(sd, 'quant_conv', cfg.z_channels, cfg.embed_dim, 1, seed)
    _conv(sd, 'post_quant_conv', cfg.embed_dim, cfg.z_channels, 1, seed)
    emb = _uniform('quantize.embeddings', (cfg.embed_dim, cfg.n_embed), math.sqrt(3.0), seed)
    if codebook_scale is not None:
        emb = (emb * np.float32(codebook_scale)).astype(np.float32)
    sd['quantize.embeddings'] = emb
    sd['quantize.ema_cluster_size_hidden'] = np.zeros((cfg.n_embed,), np.float32)
    sd['quantize.ema_dw_hidden'] = np.zeros_like(emb)
    sd['quantize.counter'] = np.zeros((), np.int64)
    return sd


# ----------------------------------------------------------------------------- MIGT layout
def make_migt_weights(cfg: MIGTConfig = None, seed: int = 0, jitter_norm: bool = True,
                      std: float = 0.02):
    """Deterministic MIGT variable dict (numpy fp32).  Dense weights are stored
    [n_in, n_out] exactly like the reference's Conv1D (migt.py:83-87)."""
    cfg = cfg or MIGTConfig()
    d = cfg.d_model
    sd = OrderedDict()

    def dense(name, nx, nf, bias_jitter=True):
        sd[name + '.weight'] = _trunc_normal(name + '.weight', (nx, nf), std, seed)
        # reference init is zero bias; a small non-zero bias makes the parity tests see it
        sd[name + '.bias'] = (_uniform(name + '.bias', (nf,), 0.02, seed) if bias_jitter
                              else np.zeros((nf,), np.float32))

    def ln(name):
        g = np.ones((d,), np.float32)
        b = np.zeros((d,), np.float32)
        if jitter_norm:
            g = g + _uniform(name + '.gamma', (d,), 0.25, seed)
            b = b + _uniform(name + '.beta', (d,), 0.1, seed)
        sd[name + '.gamma'] = g
        sd[name + '.beta'] = b

    sd['wte.weight'] = _trunc_normal('wte.weight', (cfg.n_embeddings + 2, d), std, seed)
    sd['wpe.embeddings'] = _trunc_normal('wpe.embeddings', (256, d), std, seed)
    dense('pose_embedding.c_fc', 7, 2 * d)
    dense('pose_embedding.c_proj', 2 * d, d)
    for i in range(cfg.n_layer):
        p = f'h.{i}'
        ln(p + '.ln_1')
        dense(p + '.attn.c_attn', d, 3 * d)
        dense(p + '.attn.c_proj', d, d)
        ln(p + '.ln_2')
        dense(p + '.mlp.c_fc', d, 4 * d)
        dense(p + '.mlp.c_proj', 4 * d, d)
    ln('ln_f')
    dense('pose_criterion.pose_classifier.c_fc', d, 2 * d)
    dense('pose_criterion.pose_classifier.c_proj', 2 * d, 7)
    if cfg.use_dynamic_pose_loss:
        sd['pose_loss_weighting_criterion.pos_ori_weights'] = np.array([0.0, -3.0], dtype=np.float32)     # migt.py:113
    return sd


# ----------------------------------------------------------------------------- synthetic inputs
def synthetic_scene_batch(batch: int, views: int, image_size: int = 128, seed: int = 0):
    """SURVEY.md §8(d) synthetic inputs: low-pass filtered uint8 noise frames
    [B,S,H,W,3] and cameras [B,S,7] = (xyz ~ N(0,1), unit quaternion with w>=0)."""
    r = np.random.Generator(np.random.PCG64(1000 + seed))
    f = r.integers(0, 256, size=(batch, views, image_size, image_size, 3), dtype=np.uint8).astype(np.float32)
    for _ in range(2):   # 3x3 box filter twice, edge-replicated
        p = np.pad(f, ((0, 0), (0, 0), (1, 1), (1, 1), (0, 0)), mode='edge')
        f = sum(p[:, :, dy:dy + image_size, dx:dx + image_size] for dy in range(3) for dx in range(3)) / 9.0
    # stretch contrast back to the full range so the encoder sees image-like dynamics
    lo = f.min(axis=(2, 3, 4), keepdims=True)
    hi = f.max(axis=(2, 3, 4), keepdims=True)
    frames = np.clip((f - lo) / np.maximum(hi - lo, 1e-6) * 255.0, 0, 255).astype(np.uint8)
    u = r.random(size=(batch, views, 7, 4)).sum(-1) - 2.0          # ~N(0, 1/3)
    g = (u * math.sqrt(3.0)).astype(np.float32)
    xyz = g[..., :3]
    q = g[..., 3:] + np.float32(1e-3)
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    q = q * np.where(q[..., :1] >= 0, 1.0, -1.0)
    cameras = np.concatenate([xyz, q], -1).astype(np.float32)
    return frames, cameras
