"""Tensor-level wrappers of the training-step entry points (include/vf_hip.h, csrc/train_ops.hip).
Same rules as ops.py: raw device pointers, torch = memory + stream only, no fallback."""
import ctypes

import torch

from . import _lib
from ._lib import check
from .ops import _p, _f32, _chk, _stream


def transpose(src, rows, cols, ld_src=None, batch=1, bs_src=0, out=None, ld_dst=None):
    """dst[b][c][r] = src[b][r][c]; ``ld_dst`` > rows leaves the tail of every output row untouched
    (pass a zero-filled ``out`` to get zero padding)"""
    ld_src = cols if ld_src is None else ld_src
    ld_dst = rows if ld_dst is None else ld_dst
    if out is None:
        out = torch.empty((batch, cols, ld_dst), dtype=torch.float32, device=src.device)
    lib = _lib.load()
    if src.dtype == torch.bfloat16:                 # a saved bf16 activation: widened exactly on the way
        check(lib.vf_transpose_bf16_f32(_p(_chk(src, torch.bfloat16)), _p(out), rows, cols, ld_src, ld_dst, batch, bs_src, cols * ld_dst,
                                        _stream()), 'vf_transpose_bf16_f32')
    else:
        check(lib.vf_transpose_f32(_p(_f32(src)), _p(out), rows, cols, ld_src, ld_dst, batch, bs_src, cols * ld_dst, _stream()),
              'vf_transpose_f32')
    return out


_ws_cache = {}


def _ws(nbytes, dev, key, zero=False):
    t = _ws_cache.get((key, dev))
    if t is None or t.numel() < nbytes:
        t = (torch.zeros if zero else torch.empty)(max(nbytes, 1), dtype=torch.uint8, device=dev)
        _ws_cache[(key, dev)] = t
    return t


def colsum(x, out, M, N, ld=None, accumulate=False):
    lib = _lib.load()
    ws = _ws(int(lib.vf_colsum_workspace_bytes(N)), x.device, 'colsum')
    check(lib.vf_colsum_f32(_p(_f32(x)), _p(_f32(out)), M, N, N if ld is None else ld, 1 if accumulate else 0, _p(ws), _stream()),
          'vf_colsum_f32')
    return out


def _drop4(drop):
    """(rate, seed, site[, offset]) -> the four scalars of the C-ABI; ``offset``: the first row's (elementwise sites) / first plane's
    (attention: first scene's index times H) position in the GLOBAL batch of a data-parallel step, 0 by default"""
    rate, seed, site = drop[:3]
    return float(rate), int(seed) & 0xFFFFFFFF, int(site), (int(drop[3]) if len(drop) > 3 else 0)


def layernorm_bwd(dy, x, gamma, dgamma, dbeta, rows, d, eps=1e-5, accumulate=True, res=None, also_bf16=False, drop=(0.0, 0, 0)):
    """``res``: the residual branch's gradient, added to dx in the same pass (dx = LN'(dy) + res); ``also_bf16``: returns (dx, bf16 copy);
    ``drop`` = (rate, seed, site) with ``also_bf16``: the bf16 copy is dropout_add(dx) of that site — the dY of a layer whose output went
    through that dropout (the fp32 dx, the residual path's gradient, is not masked)"""
    lib = _lib.load()
    dx = torch.empty_like(x)
    dx16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if also_bf16 else None
    if drop[0] and not also_bf16:
        raise _lib.VfError('layernorm_bwd: drop masks the bf16 copy (also_bf16=True)')
    ws = _ws(int(lib.vf_layernorm_bwd_workspace_bytes(rows, d)), x.device, 'lnbwd')
    check(lib.vf_layernorm_bwd_f32(_p(_f32(dy)), _p(_f32(x)), _p(_f32(gamma)), _p(dx), _p(dgamma), _p(dbeta), rows, d, eps,
                                   1 if accumulate else 0, _p(_f32(res)) if res is not None else None, _p(dx16), *_drop4(drop), _p(ws), _stream()),
          'vf_layernorm_bwd_f32')
    return (dx, dx16) if also_bf16 else dx


def gelu(u, out_bf16=False):
    if out_bf16:
        f = torch.empty(u.shape, dtype=torch.bfloat16, device=u.device)
        check(_lib.load().vf_gelu_bf16out_f32(_p(_f32(u)), _p(f), u.numel(), _stream()), 'vf_gelu_bf16out_f32')
        return f
    f = torch.empty_like(u)
    check(_lib.load().vf_gelu_f32(_p(_f32(u)), _p(f), u.numel(), _stream()), 'vf_gelu_f32')
    return f


def gelu_bwd(u, df, out_bf16=False):
    if out_bf16:
        du = torch.empty(u.shape, dtype=torch.bfloat16, device=u.device)
        check(_lib.load().vf_gelu_bwd_bf16out_f32(_p(_f32(u)), _p(_f32(df)), _p(du), u.numel(), _stream()), 'vf_gelu_bwd_bf16out_f32')
        return du
    du = torch.empty_like(u)
    check(_lib.load().vf_gelu_bwd_f32(_p(_f32(u)), _p(_f32(df)), _p(du), u.numel(), _stream()), 'vf_gelu_bwd_f32')
    return du


def softmax_mask_(s, batch, T, L, mask_spec, scale=1.0):
    check(_lib.load().vf_softmax_mask_f32(_p(_f32(s)), batch, T, L, mask_spec, scale, _stream()), 'vf_softmax_mask_f32')
    return s


def softmax_mask_bwd_(p, dp, batch, T, L, mask_spec, scale=1.0):
    check(_lib.load().vf_softmax_mask_bwd_f32(_p(_f32(p)), _p(_f32(dp)), batch, T, L, mask_spec, scale, _stream()),
          'vf_softmax_mask_bwd_f32')
    return dp


def softmax_ce(logits, target_i32, row_weight, rows, V, label_smoothing=0.0):
    loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
    dl = torch.empty((rows, V), dtype=torch.float32, device=logits.device)
    check(_lib.load().vf_softmax_ce_f32(_p(_f32(logits)), _p(_chk(target_i32, torch.int32)), _p(_f32(row_weight)), _p(loss), _p(dl),
                                        rows, V, float(label_smoothing), _stream()), 'vf_softmax_ce_f32')
    return loss, dl


def pose_mse(raw, gt, row_weight, rows, L, position_multiplier, w_ori=None, xyz_div=None):
    """``row_weight`` weights the position term's gradient, ``w_ori`` (default: the same) the orientation term's; ``xyz_div`` [rows]:
    per-row divisor of the predicted position (the random pose multiplier)"""
    dev = raw.device
    pos = torch.empty(rows, dtype=torch.float32, device=dev)
    ori = torch.empty(rows, dtype=torch.float32, device=dev)
    draw = torch.empty((rows, 7), dtype=torch.float32, device=dev)
    w_ori = row_weight if w_ori is None else w_ori
    check(_lib.load().vf_pose_mse_f32(_p(_f32(raw)), _p(_f32(gt)), _p(_f32(row_weight)), _p(_f32(w_ori)),
                                      _p(_f32(xyz_div)) if xyz_div is not None else None, _p(pos), _p(ori), _p(draw), rows, L,
                                      position_multiplier, _stream()), 'vf_pose_mse_f32')
    return pos, ori, draw


def embed_bwd(dh, ids_i32, dwte, dwpe, BS, L, d, vocab):
    lib = _lib.load()
    dadd = torch.empty((BS, d), dtype=torch.float32, device=dh.device)
    ws = torch.empty(int(lib.vf_embed_bwd_workspace_bytes(d, vocab)) // 4, dtype=torch.float32, device=dh.device)
    check(lib.vf_embed_bwd_f32(_p(_f32(dh)), _p(_chk(ids_i32, torch.int32)), _p(_f32(dwte)), _p(_f32(dwpe)), _p(dadd), BS, L,
                               d, vocab, _p(ws), _stream()), 'vf_embed_bwd_f32')
    return dadd


def dense_small_k_bwd(x, dy, dW, db, rows, K, N):
    check(_lib.load().vf_dense_small_k_bwd_f32(_p(_f32(x)), _p(_f32(dy)), _p(_f32(dW)), _p(_f32(db)), rows, K, N, _stream()),
          'vf_dense_small_k_bwd_f32')


def dense_small_n_supported(K, N):
    """shapes vf_dense_small_n_f32 / _wgrad_f32 take: a dense layer with at most 8 outputs (the pose head's 1536 -> 7)"""
    return N <= 8 and K % 4 == 0 and K <= 2048


def dense_small_n(x, W, b, rows, K, N):
    """out[rows][N] = x[rows][K] @ W[K][N] + b for N <= 8: one pass over x (csrc/train_ops.hip)"""
    out = torch.empty((rows, N), dtype=torch.float32, device=x.device)
    check(_lib.load().vf_dense_small_n_f32(_p(_f32(x)), _p(_f32(W)), _p(_f32(b)) if b is not None else None, _p(out), rows, K, N, x.stride(0), _stream()),
          'vf_dense_small_n_f32')
    return out


def dense_small_n_wgrad(x, dy, dW, rows, K, N, accumulate=True):
    """dW[K][N] (+)= x^T @ dy for N <= 8 (slab partial sums folded in slab order: deterministic)"""
    lib = _lib.load()
    ws = torch.empty((int(lib.vf_dense_small_n_wgrad_slabs(rows)), K * N), dtype=torch.float32, device=x.device)
    check(lib.vf_dense_small_n_wgrad_f32(_p(_f32(x)), _p(_f32(dy)), _p(_f32(dW)), _p(ws), rows, K, N, x.stride(0), 1 if accumulate else 0, _stream()),
          'vf_dense_small_n_wgrad_f32')
    return dW


def adamw_(param, grad, m, v, lr_decay, lr_adam, beta1, beta2, eps):
    check(_lib.load().vf_adamw_f32(_p(_f32(param)), _p(_f32(grad)), _p(_f32(m)), _p(_f32(v)), param.numel(), lr_decay, lr_adam,
                                   beta1, beta2, eps, _stream()), 'vf_adamw_f32')


def adamw_flat_(param, grad, m, v, nodecay_ranges, lr_decay, lr_adam, beta1, beta2, eps):
    """AdamWeightDecay over a whole flat buffer in one launch; ``nodecay_ranges`` = device int64 [R][2] of sorted element ranges that
    skip the decay (bias tensors)"""
    check(_lib.load().vf_adamw_flat_f32(_p(_f32(param)), _p(_f32(grad)), _p(_f32(m)), _p(_f32(v)), param.numel(),
                                        _p(nodecay_ranges) if nodecay_ranges is not None and nodecay_ranges.numel() else None,
                                        0 if nodecay_ranges is None else nodecay_ranges.shape[0], lr_decay, lr_adam, beta1, beta2, eps,
                                        _stream()), 'vf_adamw_flat_f32')


def adamw_pack_table(flat_param, items, nodecay_ranges=None):
    """descriptor table of vf_adamw_flat_pack_f32.  ``items`` = [(w, packed_kn or None, packed_nk or None)] with every ``w`` a contiguous 2-D fp32
    VIEW into ``flat_param`` (rows % 128 == 0, cols % 128 == 0), ``packed_kn`` the bf16 buffer ops.pack_dense_kn_bf16(w) fills, ``packed_nk`` the
    one ops.pack_dense_nk_bf16(w) fills; ``nodecay_ranges`` = the [R][2] element ranges adamw_flat_pack_ will be given (a matrix inside one
    carries the flag in its descriptor).  Returns (device table, n) after the library validated the host copy; None when a shape does not tile."""
    import numpy as np
    lib = _lib.load()
    dt = np.dtype([('offset', '<i8'), ('dst_kn', '<u8'), ('dst_nk', '<u8'), ('rows', '<i4'), ('cols', '<i4'), ('nodecay', '<i4'), ('reserved', '<i4')])
    assert dt.itemsize == _lib.ADAMW_PACK_DESC_BYTES
    base, rows_ = _f32(flat_param).data_ptr(), []
    nd = [] if nodecay_ranges is None else [tuple(int(x) for x in r) for r in nodecay_ranges.cpu().tolist()]
    for w, kn, nk in items:
        _f32(w)
        assert w.dim() == 2 and w.is_contiguous() and (w.data_ptr() - base) % 4 == 0
        off = (w.data_ptr() - base) // 4
        assert 0 <= off and off + w.numel() <= flat_param.numel()
        for buf, (K, N) in ((kn, w.shape), (nk, (w.shape[1], w.shape[0]))):
            if buf is not None:
                _chk(buf, torch.bfloat16)
                assert buf.numel() >= int(lib.vf_gemm_bf16_packed_elems(K, N))
        inside = [a <= off and off + w.numel() <= b for a, b in nd]
        assert all(i or b <= off or off + w.numel() <= a for i, (a, b) in zip(inside, nd)), 'a no-decay range cuts a weight matrix'
        rows_.append((off, 0 if kn is None else kn.data_ptr(), 0 if nk is None else nk.data_ptr(), w.shape[0], w.shape[1], int(any(inside)), 0))
    a = np.array(sorted(rows_), dtype=dt)
    rc = lib.vf_adamw_pack_check(a.ctypes.data, len(a), flat_param.numel())
    if rc == -2:                                                   # VF_ERR_UNSUPPORTED: a shape that does not tile
        return None
    check(rc, 'vf_adamw_pack_check')
    return torch.from_numpy(a.view(np.uint8).copy()).to(flat_param.device), len(a)


def adamw_flat_pack_(param, grad, m, v, nodecay_ranges, lr_decay, lr_adam, beta1, beta2, eps, table):
    """adamw_flat_ with the bf16 re-packing of the table's weight matrices in the same pass (bit-identical to adamw_flat_ + ops.pack_bf16_multi)"""
    descs, n = table
    check(_lib.load().vf_adamw_flat_pack_f32(_p(_f32(param)), _p(_f32(grad)), _p(_f32(m)), _p(_f32(v)), param.numel(),
                                             _p(nodecay_ranges) if nodecay_ranges is not None and nodecay_ranges.numel() else None,
                                             0 if nodecay_ranges is None else nodecay_ranges.shape[0], lr_decay, lr_adam, beta1, beta2, eps,
                                             descs.data_ptr(), n, _stream()), 'vf_adamw_flat_pack_f32')


def add_(a, b):
    check(_lib.load().vf_add_inplace_f32(_p(_f32(a)), _p(_f32(b)), a.numel(), _stream()), 'vf_add_inplace_f32')
    return a


def axpby(a, x, b=0.0, y=None, out=None):
    """out = a*x + b*y"""
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().vf_axpby_f32(float(a), _p(_f32(x)), float(b), _p(_f32(y)) if y is not None else None, _p(out), x.numel(),
                                   _stream()), 'vf_axpby_f32')
    return out


def clip_by_norm_(x, clip, scratch1):
    check(_lib.load().vf_clip_by_norm_f32(_p(_f32(x)), x.numel(), clip, _p(scratch1), _stream()), 'vf_clip_by_norm_f32')
    return x


def clip_grad_norm_(x, max_norm, scratch1):
    """global-norm clip of a flat gradient buffer (torch.nn.utils.clip_grad_norm_ semantics)"""
    check(_lib.load().vf_clip_grad_norm_f32(_p(_f32(x)), x.numel(), max_norm, _p(scratch1), _stream()), 'vf_clip_grad_norm_f32')
    return x


def attn_fwd_lse(q, k, v, out, B, H, T, L, ldq, ldk, ldv, ldo, scale=1.0, mask_spec=-1, drop=(0.0, 0, 0)):
    """forward attention that also returns the per-query log-sum-exp [B,H,T] the flash backward needs;
    ``drop`` = (rate, seed, site) applies attn_dropout to softmax(w) with the counter-based mask of vf_common.h"""
    lib = _lib.load()
    lse = torch.empty((B, H, T), dtype=torch.float32, device=q.device)
    check(lib.vf_attn_blockcausal_lse_f32(_p(q), _p(k), _p(v), _p(out), _p(lse), B, H, T, L, ldq, ldk, ldv, ldo, scale, 1,
                                          mask_spec, *_drop4(drop), _stream()),
          'vf_attn_blockcausal_lse_f32')
    return lse


def attn_bwd(q, k, v, out, dout, lse, dq, dk, dv, B, H, T, L, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, scale=1.0,
             mask_spec=-1, drop=(0.0, 0, 0)):
    """dQ, dK, dV of the block-causal / streams attention (written in place; the tensors may be column views)"""
    lib = _lib.load()
    D = torch.empty((B, H, T), dtype=torch.float32, device=q.device)
    check(lib.vf_attn_bwd_prep_f32(_p(dout), _p(out), _p(D), B, H, T, lddo, ldo, _stream()), 'vf_attn_bwd_prep_f32')
    check(lib.vf_attn_bwd_f32(_p(q), _p(k), _p(v), _p(dout), _p(lse), _p(D), _p(dq), _p(dk), _p(dv), B, H, T, L, ldq, ldk, ldv,
                              lddo, lddq, lddk, lddv, scale, mask_spec, *_drop4(drop), _stream()), 'vf_attn_bwd_f32')


def attn_bf16_supported(T, L):
    """shapes the bf16 training attention takes (64-token views, whole views, at most 64 of them); otherwise the f32 kernels"""
    return L == 64 and T % 64 == 0 and T // 64 <= 64


def attn_fwd_lse_bf16(q, k, v, out, B, H, T, L, ldq, ldk, ldv, ldo, scale=1.0, mask_spec=-1, drop=(0.0, 0, 0)):
    """bf16 q / k / v -> bf16 ``out`` (LDS-DMA kernel of the inference arm) plus the per-query log-sum-exp [B,H,T] fp32; ``drop`` =
    (rate, seed, site): attn_dropout on softmax(w) with the masks of vf_common.h (the same masks as attn_fwd_lse)"""
    for t in (q, k, v, out):
        _chk(t, torch.bfloat16)
    lse = torch.empty((B, H, T), dtype=torch.float32, device=q.device)
    check(_lib.load().vf_attn_blockcausal_bf16_lse(_p(q), _p(k), _p(v), _p(out), _p(lse), B, H, T, L, ldq, ldk, ldv, ldo, scale, mask_spec,
                                                   *_drop4(drop), _stream()), 'vf_attn_blockcausal_bf16_lse')
    return lse


def attn_bwd_bf16(q, k, v, out, dout, lse, dq, dk, dv, B, H, T, L, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, scale=1.0, mask_spec=-1,
                  drop=(0.0, 0, 0), kv_stream=None):
    """dQ, dK, dV (fp32, or bf16 when the output tensors are bf16; written in place; column views allowed) from bf16 q / k / v / out / dout
    on the bf16 matrix pipe; ``drop`` as in the forward (the masks are recomputed).  ``kv_stream``: the dK / dV launch goes to that stream
    (after the row sums D on the current one) and the current stream waits for it at the end — the two launches are independent and each ends
    on a partly filled round of its heaviest workgroups; same results."""
    lib = _lib.load()
    for t in (q, k, v, out, dout):
        _chk(t, torch.bfloat16)
    D = torch.empty((B, H, T), dtype=torch.float32, device=q.device)
    check(lib.vf_attn_bwd_prep_bf16(_p(dout), _p(out), _p(D), B, H, T, lddo, ldo, _stream()), 'vf_attn_bwd_prep_bf16')
    o16 = dq.dtype == torch.bfloat16
    for t in (dq, dk, dv):
        _chk(t, torch.bfloat16 if o16 else torch.float32)

    def launch(dq_, dk_, dv_):
        check(lib.vf_attn_bwd_bf16(_p(q), _p(k), _p(v), _p(dout), _p(lse), _p(D), _p(dq_) if dq_ is not None else None,
                                   _p(dk_) if dk_ is not None else None, _p(dv_) if dv_ is not None else None, 1 if o16 else 0, B, H, T, L, ldq, ldk,
                                   ldv, lddo, lddq, lddk, lddv, scale, mask_spec, *_drop4(drop), _stream()), 'vf_attn_bwd_bf16')
    if kv_stream is None:
        return launch(dq, dk, dv)
    main = torch.cuda.current_stream(q.device)
    kv_stream.wait_stream(main)                               # q, k, v, dout, lse, D are complete
    with torch.cuda.stream(kv_stream):
        launch(None, dk, dv)
    launch(dq, None, None)
    for t in (q, k, v, dout, lse, D, dk, dv):
        t.record_stream(kv_stream)
    main.wait_stream(kv_stream)


def dropout_add(x, rate, seed, site, res=None, out=None, cols=None, row0=0):
    """out = keep ? x / (1 - rate) : 0 [+ res] with the counter-based mask of (seed, site) over x viewed as [rows][cols] (``cols`` defaults
    to the last dimension: mask group ((m + row0) >> 2) * cols + n, position (m + row0) & 3 — csrc/vf_common.h; ``row0``: the first row's
    index in the global batch of a data-parallel step); in place when out is x"""
    if out is None:
        out = torch.empty_like(x)
    cols = int(x.shape[-1]) if cols is None else int(cols)
    if cols <= 0 or x.numel() % cols:
        raise _lib.VfError('dropout_add: cols must divide the element count')
    check(_lib.load().vf_dropout_add_f32(_p(_f32(x)), _p(_f32(res)) if res is not None else None, _p(out), x.numel() // cols, cols, int(row0),
                                         float(rate), int(seed) & 0xFFFFFFFF, int(site), _stream()), 'vf_dropout_add_f32')
    return out


# ------------------------------------------------------------------ VQGAN backward helpers (csrc/vqgan_bwd.hip)
def gather_transpose(src, n_img, Hin, Win, C, Hout, Wout, stride=1, oy=0, ox=0, out=None):
    """[C][n_img*Hout*Wout]: channel-major, tap-shifted view of an NHWC activation (zero outside the image)"""
    P = n_img * Hout * Wout
    dst = torch.empty((C, P), dtype=torch.float32, device=src.device) if out is None else _f32(out)
    check(_lib.load().vf_gather_transpose_f32(_p(_f32(src)), _p(dst), n_img, Hin, Win, C, Hout, Wout, stride, oy, ox, P, _stream()),
          'vf_gather_transpose_f32')
    return dst


def upsample2_bwd(du, n_img, H, W, C):
    dx = torch.empty((n_img * H * W, C), dtype=torch.float32, device=du.device)
    check(_lib.load().vf_upsample2_bwd_f32(_p(_f32(du)), _p(dx), n_img, H, W, C, _stream()), 'vf_upsample2_bwd_f32')
    return dx


def groupnorm_bwd(x, da, mean_c, scale_c, gamma, beta, n_img, HW, C, swish, groups=32):
    """-> (dx, dgamma [C], dbeta [C]) of a = swish?(GroupNorm(x))"""
    lib = _lib.load()
    dx = torch.empty_like(x)
    chan = torch.empty((n_img, C, 2), dtype=torch.float32, device=x.device)
    ws = _ws(int(lib.vf_groupnorm_bwd_workspace_bytes(n_img, HW, C, groups)), x.device, 'gnbwd')
    check(lib.vf_groupnorm_bwd_f32(_p(_f32(x)), _p(_f32(da)), _p(mean_c), _p(scale_c), _p(_f32(gamma)), _p(_f32(beta)), _p(dx),
                                   _p(chan), n_img, HW, C, groups, 1 if swish else 0, 0, _p(ws), _stream()), 'vf_groupnorm_bwd_f32')
    sums = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    colsum(chan.view(n_img, 2 * C), sums, n_img, 2 * C)
    sums = sums.view(C, 2)
    return dx, sums[:, 0].contiguous(), sums[:, 1].contiguous()


def softmax_rows_bwd_(p, dp, rows, n, scale):
    check(_lib.load().vf_softmax_rows_bwd_f32(_p(_f32(p)), _p(_f32(dp)), rows, n, float(scale), _stream()), 'vf_softmax_rows_bwd_f32')
    return dp


def l1_loss(x, y, grad_weight):
    """-> (sum |y - x| as a 0-d tensor, dy = sign(y - x) * grad_weight)"""
    lib = _lib.load()
    n = x.numel()
    dy = torch.empty_like(y)
    part = torch.empty(int(lib.vf_l1_loss_partials(n)), dtype=torch.float32, device=x.device)
    check(lib.vf_l1_loss_f32(_p(_f32(x)), _p(_f32(y)), _p(dy), _p(part), n, float(grad_weight), _stream()), 'vf_l1_loss_f32')
    total = torch.empty(1, dtype=torch.float32, device=x.device)
    colsum(part.view(-1, 1), total, part.numel(), 1)
    return total[0], dy


# ------------------------------------------------------------------ perceptual loss pieces (csrc/lpips.hip)
def lpips_scaling(x, shift3, scale3, backward=False):
    """ScalingLayer of lpips on NHWC [.., 3] rows: (x - shift) / scale; backward: x / scale"""
    import ctypes
    y = torch.empty_like(x)
    sh = (ctypes.c_float * 3)(*[float(v) for v in shift3])
    sc = (ctypes.c_float * 3)(*[float(v) for v in scale3])
    check(_lib.load().vf_lpips_scaling_f32(_p(_f32(x)), _p(y), x.numel() // 3, sh, sc, 1 if backward else 0, _stream()),
          'vf_lpips_scaling_f32')
    return y


def relu_(x):
    check(_lib.load().vf_relu_f32(_p(_f32(x)), x.numel(), _stream()), 'vf_relu_f32')
    return x


def relu_bwd_(dy, y):
    check(_lib.load().vf_relu_bwd_f32(_p(_f32(dy)), _p(_f32(y)), dy.numel(), _stream()), 'vf_relu_bwd_f32')
    return dy


def maxpool2(x, n_img, Hout, Wout, C):
    y = torch.empty((n_img * Hout * Wout, C), dtype=torch.float32, device=x.device)
    check(_lib.load().vf_maxpool2_f32(_p(_f32(x)), _p(y), n_img, Hout, Wout, C, _stream()), 'vf_maxpool2_f32')
    return y


def maxpool2_bwd(x, dy, n_img, Hout, Wout, C):
    dx = torch.empty((n_img * Hout * Wout * 4, C), dtype=torch.float32, device=x.device)
    check(_lib.load().vf_maxpool2_bwd_f32(_p(_f32(x)), _p(_f32(dy)), _p(dx), n_img, Hout, Wout, C, _stream()), 'vf_maxpool2_bwd_f32')
    return dx


def lpips_head(f0, f1, w, n_img, HW, C):
    """-> per-image sums over pixels of sum_c w_c (f0n - f1n)^2, [n_img]"""
    lib = _lib.load()
    nb = int(lib.vf_lpips_head_blocks(HW))
    part = torch.empty((nb, n_img), dtype=torch.float32, device=f0.device)
    check(lib.vf_lpips_head_f32(_p(_f32(f0)), _p(_f32(f1)), _p(_f32(w)), _p(part), n_img, HW, C, _stream()), 'vf_lpips_head_f32')
    sums = torch.empty(n_img, dtype=torch.float32, device=f0.device)
    colsum(part, sums, nb, n_img)
    return sums


def lpips_head_bwd(f0, f1, w, df1, npix, C, gscale, accumulate):
    check(_lib.load().vf_lpips_head_bwd_f32(_p(_f32(f0)), _p(_f32(f1)), _p(_f32(w)), _p(_f32(df1)), npix, C, float(gscale),
                                            1 if accumulate else 0, _stream()), 'vf_lpips_head_bwd_f32')
    return df1


def conv3_wgrad_supported(Cin, n_img, Hout, Wout):
    pow2 = lambda v: v > 0 and (v & (v - 1)) == 0      # noqa: E731
    return Cin % 128 == 0 and pow2(Hout) and pow2(Wout) and Wout % 4 == 0 and (n_img * Hout * Wout) % 64 == 0


def conv3_wgrad(x, dy, n_img, Hin, Win, Cin, Hout, Wout, Cout, mode):
    """weight + bias gradient of a 3x3 convolution (forward mode ``mode``) -> [9*Cin + 1][Cout]: rows (ky*3+kx)*Cin + ci, last row
    = bias gradient.  x: NHWC input rows of the forward conv, dy: [P][Cout]."""
    from . import ops
    lib = _lib.load()
    P = n_img * Hout * Wout
    rows = int(lib.vf_conv3_wgrad_x6_rows(Cin))
    tiles = (9 * (Cin // 128) + 1) * ((Cout + 127) // 128)
    splits = max(1, min(256, 512 // tiles, P // 512))
    dyp = ops.pack_dense_kn_x6(dy)
    slabs = torch.empty((splits, rows, Cout), dtype=torch.float32, device=x.device)
    check(lib.vf_conv3_wgrad_x6(_p(_f32(x)), _p(dyp), _p(slabs), n_img, Hin, Win, Cin, Hout, Wout, Cout, mode, splits, _stream()),
          'vf_conv3_wgrad_x6')
    if splits == 1:
        return slabs[0]
    n = rows * Cout
    if n % 4:
        raise _lib.VfError('conv3_wgrad: (9*Cin+1)*Cout must be a multiple of 4 (vf_sum_slabs_f32 works in float4)')
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    check(lib.vf_sum_slabs_f32(_p(slabs), splits, n, n, _p(out), 0, _stream()), 'vf_sum_slabs_f32')
    return out.view(rows, Cout)
