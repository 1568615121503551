"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the
reference's VQGAN codebook model.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product path (``viewformer_amd``) never does.

Every function is a pure function of ``(state_dict, config, inputs)`` and restates,
with torch *CPU* functional ops, the graph of viewformer/models/vqgan_th.py and
viewformer/models/utils_th.py.  ``dtype=torch.float64`` gives the high-precision
arm used to measure the fp32 error of both this oracle and the HIP path.

Pinning: ``tests/golden/make_golden.py`` imports the reference itself (in the build
container, with stub modules for aparse / pytorch_lightning / lpips), loads the
same synthetic state dict and records codes / z / pixels; ``tests/test_oracle_vqgan.py``
checks this restatement against those vectors (codes bit-exact, floats <= 1e-5).
"""
import torch
import torch.nn.functional as F


def _t(sd, name, dtype):
    v = sd[name]
    if not isinstance(v, torch.Tensor):
        v = torch.from_numpy(v)
    return v.to(dtype)


def swish(x):
    """vqgan_th.py:11-13"""
    return x * torch.sigmoid(x)


def group_norm(sd, name, x, dtype):
    """vqgan_th.py:16-17 — GroupNorm(32 groups, eps=1e-6, affine)."""
    return F.group_norm(x, 32, _t(sd, name + '.weight', dtype), _t(sd, name + '.bias', dtype), eps=1e-6)


def conv(sd, name, x, dtype, stride=1, padding=0):
    return F.conv2d(x, _t(sd, name + '.weight', dtype), _t(sd, name + '.bias', dtype), stride=stride, padding=padding)


def resnet_block(sd, name, x, cin, cout, dtype):
    """vqgan_th.py:78-90"""
    h = conv(sd, name + '.conv1', swish(group_norm(sd, name + '.norm1', x, dtype)), dtype, padding=1)
    h = conv(sd, name + '.conv2', swish(group_norm(sd, name + '.norm2', h, dtype)), dtype, padding=1)
    if cin != cout:
        x = conv(sd, name + '.nin_shortcut', x, dtype)
    return x + h


def attn_block(sd, name, x, dtype):
    """vqgan_th.py:120-144 — single head, scale C^-0.5, softmax over keys."""
    h_ = group_norm(sd, name + '.norm', x, dtype)
    q = conv(sd, name + '.q', h_, dtype)
    k = conv(sd, name + '.k', h_, dtype)
    v = conv(sd, name + '.v', h_, dtype)
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + conv(sd, name + '.proj_out', h_, dtype)


def downsample(sd, name, x, dtype):
    """vqgan_th.py:45-49 — zero pad (right, bottom) then 3x3 stride 2."""
    return conv(sd, name, F.pad(x, (0, 1, 0, 1), mode='constant', value=0), dtype, stride=2)


def upsample(sd, name, x, dtype):
    """vqgan_th.py:29-32 — nearest x2 then 3x3 pad 1."""
    return conv(sd, name, F.interpolate(x, scale_factor=2.0, mode='nearest'), dtype, padding=1)


def encoder(sd, cfg, x, dtype=torch.float32, taps=None):
    """Encoder.forward, vqgan_th.py:203-225 (constructor :147-201)."""
    x = x.to(dtype)
    ch, mult, nrb = cfg.ch, list(cfg.ch_mult), cfg.num_res_blocks
    nres = len(mult)
    in_mult = [1] + mult
    res = cfg.image_size
    h = conv(sd, 'encoder.conv_in', x, dtype, padding=1)
    if taps is not None:
        taps['encoder.conv_in'] = h
    for lvl in range(nres):
        bin_, bout = ch * in_mult[lvl], ch * mult[lvl]
        for b in range(nrb):
            h = resnet_block(sd, f'encoder.down.{lvl}.block.{b}', h, bin_, bout, dtype)
            bin_ = bout
            if res in cfg.attn_resolutions:
                h = attn_block(sd, f'encoder.down.{lvl}.attn.{b}', h, dtype)
            if taps is not None:
                taps[f'encoder.down.{lvl}.{b}'] = h
        if lvl != nres - 1:
            h = downsample(sd, f'encoder.down.{lvl}.downsample.conv', h, dtype)
            res //= 2
    h = resnet_block(sd, 'encoder.mid.block_1', h, bin_, bin_, dtype)
    h = attn_block(sd, 'encoder.mid.attn_1', h, dtype)
    h = resnet_block(sd, 'encoder.mid.block_2', h, bin_, bin_, dtype)
    h = swish(group_norm(sd, 'encoder.norm_out', h, dtype))
    return conv(sd, 'encoder.conv_out', h, dtype, padding=1)


def decoder(sd, cfg, z, dtype=torch.float32, taps=None):
    """Decoder.forward, vqgan_th.py:291-318 (constructor :228-289)."""
    ch, mult, nrb = cfg.ch, list(cfg.ch_mult), cfg.num_res_blocks
    nres = len(mult)
    bin_ = ch * mult[nres - 1]
    res = cfg.image_size // 2 ** (nres - 1)
    h = conv(sd, 'decoder.conv_in', z.to(dtype), dtype, padding=1)
    h = resnet_block(sd, 'decoder.mid.block_1', h, bin_, bin_, dtype)
    h = attn_block(sd, 'decoder.mid.attn_1', h, dtype)
    h = resnet_block(sd, 'decoder.mid.block_2', h, bin_, bin_, dtype)
    for lvl in reversed(range(nres)):
        bout = ch * mult[lvl]
        for b in range(nrb + 1):
            h = resnet_block(sd, f'decoder.up.{lvl}.block.{b}', h, bin_, bout, dtype)
            bin_ = bout
            if res in cfg.attn_resolutions:
                h = attn_block(sd, f'decoder.up.{lvl}.attn.{b}', h, dtype)
        if taps is not None:
            taps[f'decoder.up.{lvl}'] = h
        if lvl != 0:
            h = upsample(sd, f'decoder.up.{lvl}.upsample.conv', h, dtype)
            res *= 2
    h = swish(group_norm(sd, 'decoder.norm_out', h, dtype))
    return conv(sd, 'decoder.conv_out', h, dtype, padding=1)


def quantize_distances(sd, z_nchw, dtype=torch.float32):
    """utils_th.py:34-40 — expanded-form squared distances [N*h*w, K]."""
    emb = _t(sd, 'quantize.embeddings', dtype)
    x = z_nchw.to(dtype).permute(0, 2, 3, 1)
    flatten = x.reshape(-1, x.size(-1))
    dist = (flatten.pow(2).sum(1, keepdim=True)
            - 2 * flatten @ emb
            + emb.pow(2).sum(0, keepdim=True))
    return dist


def embed_code(sd, codes, dtype=torch.float32):
    """utils_th.py:70-72"""
    emb = _t(sd, 'quantize.embeddings', dtype)
    return F.embedding(codes.long(), emb.transpose(0, 1)).permute(0, 3, 1, 2).contiguous()


def quantize(sd, z_nchw, dtype=torch.float32):
    """QuantizeEMA.forward eval branch, utils_th.py:32-44,66-68.
    Returns (quantize, diff, embed_ind[int64 N,h,w]); ties -> lowest index
    (torch CPU ``max`` returns the first maximum)."""
    dist = quantize_distances(sd, z_nchw, dtype)
    _, ind = (-dist).max(1)
    n, _, h, w = z_nchw.shape
    ind = ind.view(n, h, w)
    q = embed_code(sd, ind, dtype)
    zin = z_nchw.to(dtype)
    diff = (q - zin).pow(2).mean()
    q = zin + (q - zin)          # straight-through value, utils_th.py:67 (differs from q in the last bit)
    return q, diff, ind


def quantize_train_step(state, z_nchw, decay=0.99, eps=1e-5, all_reduce=None, dtype=torch.float32):
    """QuantizeEMA.forward TRAINING branch, utils_th.py:32-68.  ``state`` = dict(embeddings [D,K], ema_cluster_size_hidden [K],
    ema_dw_hidden [D,K], counter int) is updated in place exactly as the reference's buffers are: the lookup and the returned
    quantize/diff use the embeddings BEFORE the update (:41-44), then the per-code counts and sums (:47-48) are optionally
    all-reduced over replicas (:50-52), folded into the EMA buffers (:55-57) and turned into the new, bias-corrected and
    Laplace-smoothed embeddings (:59-64).  ``all_reduce(t)`` sums a tensor over replicas in place (None = single replica)."""
    emb = state['embeddings'].to(dtype)
    q, diff, ind = quantize({'quantize.embeddings': emb}, z_nchw, dtype)
    K = emb.shape[1]
    x = z_nchw.to(dtype).permute(0, 2, 3, 1)
    flatten = x.reshape(-1, x.size(-1))
    onehot = F.one_hot(ind.reshape(-1), K).to(dtype)
    counts = onehot.sum(0)                                              # :47
    embed_sum = flatten.transpose(0, 1) @ onehot                        # :48
    if all_reduce is not None:                                          # :50-52
        all_reduce(counts)
        all_reduce(embed_sum)
    cs, dw = state['ema_cluster_size_hidden'].to(dtype), state['ema_dw_hidden'].to(dtype)
    cs = cs + (counts - cs) * (1 - decay)                               # :55  (Tensor.add_(other, alpha))
    dw = dw + (embed_sum - dw) * (1 - decay)                            # :56
    counter = int(state['counter']) + 1                                 # :57
    corr = 1.0 - torch.pow(torch.tensor(decay, dtype=dtype), counter)   # bias correction of the properties :24-30
    ema_cs, ema_dw = cs / corr, dw / corr
    n = ema_cs.sum()                                                    # :59
    cluster = (ema_cs + eps) / (n + K * eps) * n                        # :60-62
    state.update(embeddings=ema_dw / cluster.unsqueeze(0), ema_cluster_size_hidden=cs, ema_dw_hidden=dw, counter=counter)   # :63-64
    return q, diff, ind


def encode_z(sd, cfg, x, dtype=torch.float32):
    """encoder + quant_conv (vqgan_th.py:380-381): the vectors fed to the lookup."""
    return conv(sd, 'quant_conv', encoder(sd, cfg, x, dtype), dtype)


def encode(sd, cfg, x, dtype=torch.float32):
    """VQGAN.encode, vqgan_th.py:379-383 -> (quant, emb_loss, codes)."""
    with torch.no_grad():
        return quantize(sd, encode_z(sd, cfg, x, dtype), dtype)


def decode(sd, cfg, quant, dtype=torch.float32):
    """VQGAN.decode, vqgan_th.py:385-388."""
    with torch.no_grad():
        return decoder(sd, cfg, conv(sd, 'post_quant_conv', quant.to(dtype), dtype), dtype)


def decode_code(sd, cfg, codes, dtype=torch.float32):
    """VQGAN.decode_code, vqgan_th.py:390-393."""
    return decode(sd, cfg, embed_code(sd, codes, dtype), dtype)


# ---- image pre/post-processing used by the evaluators (TF semantics restated) ----------
def resize_u8(images_u8_nhwc, image_size, method=None):
    """``resize`` / ``resize_th`` (viewformer/data/_common.py:19-61) restated in numpy fp32: uint8 -> /255 -> torch interpolate
    ('nearest' when enlarging, bilinear align_corners=False when shrinking) -> clamp -> *255 -> truncating uint8 cast.  The CPU
    interpolation of torch evaluates fma(l0, a, l1*b) per axis (x, then y); emulated with an fp64 product-sum rounded once.
    Pinned by tests/golden/resize.npz (recorded from the reference)."""
    import numpy as np
    img = np.asarray(images_u8_nhwc)
    n, H, W, C = img.shape
    if H == image_size:
        return img
    if method is None:
        method = 'nearest' if image_size > H else 'bilinear'
    f32 = np.float32
    p = (img.astype(np.float32) / f32(255.0)).astype(np.float32)
    if method == 'nearest':
        def idx(inp):
            scale = f32(inp) / f32(image_size)
            return np.minimum(np.floor(np.arange(image_size, dtype=np.float32) * scale).astype(np.int64), inp - 1)
        v = p[:, idx(H)][:, :, idx(W)]
    else:
        def axis(inp):
            scale = f32(inp) / f32(image_size)
            src = np.maximum(scale * (np.arange(image_size, dtype=np.float32) + f32(0.5)) - f32(0.5), f32(0))
            i0 = src.astype(np.int64)
            l1 = (src - i0.astype(np.float32)).astype(np.float32)
            return i0, i0 + (i0 < inp - 1), (f32(1) - l1).astype(np.float32), l1

        def fma(l0, a, l1, b):
            t = (l1 * b).astype(np.float32)
            return (l0.astype(np.float64) * a.astype(np.float64) + t.astype(np.float64)).astype(np.float32)
        y0, y1, ly0, ly1 = axis(H)
        x0, x1, lx0, lx1 = axis(W)
        lx0, lx1 = lx0[None, None, :, None], lx1[None, None, :, None]
        top = fma(lx0, p[:, y0][:, :, x0], lx1, p[:, y0][:, :, x1])
        bot = fma(lx0, p[:, y1][:, :, x0], lx1, p[:, y1][:, :, x1])
        v = fma(ly0[None, :, None, None], top, ly1[None, :, None, None], bot)
    return (np.clip(v, 0, 1) * f32(255.0)).astype(np.float32).astype(np.uint8)


def preprocess_u8(images_u8_nhwc):
    """evaluate_transformer.py:105-108: tf.image.convert_image_dtype(uint8->float32)
    multiplies by fp32(1/255) (tensorflow==2.4.1, third-party), then ``* 2 - 1``.
    Returns NCHW fp32 (the Torch calling convention)."""
    x = images_u8_nhwc.to(torch.float32) * torch.tensor(1.0 / 255, dtype=torch.float32)
    x = x * 2 - 1
    return x.permute(0, 3, 1, 2).contiguous()


def postprocess_u8(dec_nchw):
    """evaluate_transformer.py:128-129: clip[-1,1] -> /2+0.5 -> convert_image_dtype(uint8)
    = truncating cast of x * 255.5 (tensorflow==2.4.1 image_ops_impl, saturate=False)."""
    x = dec_nchw.to(torch.float32).clamp(-1, 1) / 2 + 0.5
    x = (x * torch.tensor(255.5, dtype=torch.float32)).to(torch.int32).clamp(0, 255).to(torch.uint8)
    return x.permute(0, 2, 3, 1).contiguous()
