"""Thin tensor-level wrappers over the C-ABI (include/vf_hip.h).

PyTorch is used for device memory and the stream handle only: every wrapper hands raw
device pointers to libvf_hip.so and launches on torch's current HIP stream.  Nothing here
computes with torch ops, and nothing falls back to them.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import VfIgemmArgs, check

MODE_GEMM, MODE_CONV3_S1, MODE_CONV3_S2PAD, MODE_CONV3_UP2 = 0, 1, 2, 3
EPI_NONE, EPI_GELU, EPI_GELU_BWD, EPI_GELU_DUAL = 0, 1, 2, 3


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _chk(t, dtype=torch.float32, name='tensor'):
    if not t.is_cuda:
        raise _lib.VfError(f'{name} must live on the GPU (no CPU fallback for the hot path)')
    if t.dtype != dtype:
        raise TypeError(f'{name} must be {dtype}, got {t.dtype}')
    return t


def _f32(t, name='tensor'):
    return _chk(t, torch.float32, name)


# ------------------------------------------------------------------ weight packing
def packed_floats(K, N, taps=1):
    return int(_lib.load().vf_igemm_packed_floats(K, N, taps))


def pack(src, K, N, taps, sk, sn, st, batch=1, src_bstride=0, out=None):
    """pack element (tap,k,n) = src[tap*st + k*sk + n*sn] into the fragment-major layout"""
    lib = _lib.load()
    n = packed_floats(K, N, taps)
    if out is None:
        out = torch.empty(batch * n, dtype=torch.float32, device=src.device)
    check(lib.vf_igemm_pack_f32(_p(_f32(src)), _p(out), K, N, taps, sk, sn, st, batch, src_bstride, _stream()),
          'vf_igemm_pack_f32')
    return out


def pack_conv_oihw(w):
    """torch Conv2d weight [Cout][Cin][kh][kw], kh=kw in {1,3}"""
    w = _f32(w).contiguous()
    cout, cin, kh, kw = w.shape
    taps = kh * kw
    return pack(w, cin, cout, taps, sk=taps, sn=cin * taps, st=1)


def pack_dense_kn(w):
    """Conv1D weight [nx][nf] (x @ W)"""
    w = _f32(w).contiguous()
    k, n = w.shape
    return pack(w, k, n, 1, sk=n, sn=1, st=0)


def pack_dense_nk(w, n_rows=None):
    """transposed weight [N][K] (x @ W^T), optionally only the first n_rows rows"""
    w = _f32(w).contiguous()
    n, k = w.shape
    n = n if n_rows is None else n_rows
    return pack(w, k, n, 1, sk=1, sn=k, st=0)


# ------------------------------------------------------------------ implicit GEMM
def pack_dense_kn_bf16(w):
    """Conv1D weight [nx][nf] -> bf16 fragment packing for vf_gemm_bf16"""
    lib = _lib.load()
    w = _f32(w).contiguous()
    k, n = w.shape
    out = torch.empty(int(lib.vf_gemm_bf16_packed_elems(k, n)), dtype=torch.bfloat16, device=w.device)
    check(lib.vf_gemm_bf16_pack(_p(w), _p(out), k, n, n, 1, 1, 0, _stream()), 'vf_gemm_bf16_pack')
    return out


def pack_bf16_multi(items):
    """many bf16 packings in one launch.  ``items`` = [(w fp32 [rows][cols] contiguous, transposed, out)]: transposed False packs w as
    [K = rows][N = cols] (pack_dense_kn_bf16), True as the [N = rows][K = cols] operand (pack_dense_nk_bf16); ``out`` = the bf16 buffer to
    fill (vf_gemm_bf16_packed_elems(K, N)).  Returns a closure that re-runs the same launch (descriptors stay on the device)."""
    import numpy as np
    lib = _lib.load()
    dt = np.dtype([('src', '<u8'), ('dst', '<u8'), ('K', '<i4'), ('N', '<i4'), ('sk', '<i8'), ('sn', '<i8')])
    assert dt.itemsize == _lib.PACK_DESC_BYTES
    a = np.zeros(len(items), dtype=dt)
    for i, (w, tr, out) in enumerate(items):
        _f32(w); _chk(out, torch.bfloat16)
        assert w.is_contiguous() and w.dim() == 2
        r, c = w.shape
        K, N, sk, sn = (c, r, 1, c) if tr else (r, c, c, 1)
        assert out.numel() >= int(lib.vf_gemm_bf16_packed_elems(K, N))
        a[i] = (w.data_ptr(), out.data_ptr(), K, N, sk, sn)
    descs = torch.from_numpy(a.view(np.uint8).copy()).to(items[0][0].device)
    n = len(items)

    def run(first=0, count=None):
        """re-pack items [first, first + count) (default: all of them)"""
        count = n - first if count is None else count
        if count <= 0:
            return
        assert 0 <= first and first + count <= n
        check(lib.vf_gemm_bf16_pack_multi(descs.data_ptr() + first * _lib.PACK_DESC_BYTES, count, _stream()), 'vf_gemm_bf16_pack_multi')
    run()
    return run


def pack_dense_nk_bf16(w, n_rows=None):
    """transposed weight [N][K] (x @ W^T; tied LM head, 1x1 conv [Cout][Cin]) -> bf16 packing"""
    lib = _lib.load()
    w = _f32(w).contiguous()
    n, k = w.shape
    n = n if n_rows is None else n_rows
    out = torch.empty(int(lib.vf_gemm_bf16_packed_elems(k, n)), dtype=torch.bfloat16, device=w.device)
    check(lib.vf_gemm_bf16_pack(_p(w), _p(out), k, n, 1, k, 1, 0, _stream()), 'vf_gemm_bf16_pack')
    return out


def pack_conv3_bf16(w_oihw):
    lib = _lib.load()
    w = _f32(w_oihw).contiguous()
    cout, cin = w.shape[:2]
    out = torch.empty(int(lib.vf_conv3_bf16_packed_elems(cin, cout)), dtype=torch.bfloat16, device=w.device)
    check(lib.vf_conv3_bf16_pack(_p(w), _p(out), cin, cout, _stream()), 'vf_conv3_bf16_pack')
    return out


def pack_conv3_x6(w_oihw):
    """OIHW fp32 3x3 weight -> three fragment-packed bf16 planes (exact split w = h + m + l) for vf_conv3_halo_x6"""
    lib = _lib.load()
    w = _f32(w_oihw).contiguous()
    cout, cin = w.shape[:2]
    out = torch.empty(int(lib.vf_conv3_x6_packed_elems(cin, cout)), dtype=torch.bfloat16, device=w.device)
    check(lib.vf_conv3_x6_pack(_p(w), _p(out), cin, cout, _stream()), 'vf_conv3_x6_pack')
    return out


def pack_conv3_x3h(w_oihw):
    """OIHW fp32 3x3 weight -> two fragment-packed f16 planes of w * S (S a power of two, 1/S stored behind them) for
    vf_conv3_halo_x3h"""
    lib = _lib.load()
    w = _f32(w_oihw).contiguous()
    cout, cin = w.shape[:2]
    out = torch.empty(int(lib.vf_conv3_x3h_packed_elems(cin, cout)), dtype=torch.float16, device=w.device)
    check(lib.vf_conv3_x3h_pack(_p(w), _p(out), cin, cout, _stream()), 'vf_conv3_x3h_pack')
    return out


def conv3_x3h_supported(mode, Cin, Cout, Hout, Wout):
    """shape rules of vf_conv3_halo_x3h: the x6 kernel's, plus the 16x16 -> 8x8 stride-2 convolution (two images per tile)"""
    if mode == MODE_CONV3_S2PAD and Hout == 8 and Wout == 8 and Cin % 32 == 0 and Cout % 128 == 0:
        return True
    return conv3_x6_supported(mode, Cin, Cout, Hout, Wout)


def pack_dense_kn_x6(w):
    """Conv1D weight [nx][nf] -> 3-plane split packing for vf_gemm_x6"""
    lib = _lib.load()
    w = _f32(w).contiguous()
    k, n = w.shape
    out = torch.empty(int(lib.vf_gemm_x6_packed_elems(k, n)), dtype=torch.bfloat16, device=w.device)
    check(lib.vf_gemm_x6_pack(_p(w), _p(out), k, n, n, 1, _stream()), 'vf_gemm_x6_pack')
    return out


def pack_dense_nk_x6(w, n_rows=None):
    """transposed weight [N][K] (x @ W^T; tied LM head, 1x1 conv [Cout][Cin]) -> 3-plane split packing"""
    lib = _lib.load()
    w = _f32(w).contiguous()
    n, k = w.shape
    n = n if n_rows is None else n_rows
    out = torch.empty(int(lib.vf_gemm_x6_packed_elems(k, n)), dtype=torch.bfloat16, device=w.device)
    check(lib.vf_gemm_x6_pack(_p(w), _p(out), k, n, 1, k, _stream()), 'vf_gemm_x6_pack')
    return out


def pack_dense_kn_x3h(w):
    """Conv1D weight [nx][nf] -> two-plane split-fp16 packing (w * S, 1/S behind the planes) for vf_gemm_x3h"""
    lib = _lib.load()
    w = _f32(w).contiguous()
    k, n = w.shape
    out = torch.empty(int(lib.vf_gemm_x3h_packed_elems(k, n)), dtype=torch.float16, device=w.device)
    check(lib.vf_gemm_x3h_pack(_p(w), _p(out), k, n, n, 1, _stream()), 'vf_gemm_x3h_pack')
    return out


def pack_dense_nk_x3h(w, n_rows=None):
    """transposed weight [N][K] (x @ W^T; tied LM head, 1x1 conv [Cout][Cin]) -> two-plane split-fp16 packing"""
    lib = _lib.load()
    w = _f32(w).contiguous()
    n, k = w.shape
    n = n if n_rows is None else n_rows
    out = torch.empty(int(lib.vf_gemm_x3h_packed_elems(k, n)), dtype=torch.float16, device=w.device)
    check(lib.vf_gemm_x3h_pack(_p(w), _p(out), k, n, 1, k, _stream()), 'vf_gemm_x3h_pack')
    return out


def gemm_x6_splitk(x, w_packed, M, Cin, Cout, dst, splits, lda=None, accumulate=True):
    """dst[M][Cout] (+)= x @ W with the reduction split over ``splits`` workgroup groups (deterministic: slabs summed in order);
    for GEMMs whose output has too few 128x128 tiles to fill the chip but a long reduction (the training step's dW)"""
    lib = _lib.load()
    slabs = torch.empty((splits, M, Cout), dtype=torch.float32, device=x.device)
    igemm(x, w_packed, M, Cin, Cout, slabs, lda=lda, x6=True, split_k=splits, stride_out=M * Cout)
    check(lib.vf_sum_slabs_f32(_p(slabs), splits, M * Cout, M * Cout, _p(_f32(dst)), 1 if accumulate else 0, _stream()),
          'vf_sum_slabs_f32')
    return dst


def conv3_x6_supported(mode, Cin, Cout, Hout, Wout):
    """shape rules of vf_conv3_halo_x6 (host-side mirror so callers can pick the packing up front)"""
    if mode not in (MODE_CONV3_S1, MODE_CONV3_UP2, MODE_CONV3_S2PAD) or Cin % 32 or Cout % 128:
        return False
    return (Hout % 8 == 0 and Wout % 16 == 0) or (mode == MODE_CONV3_S1 and Hout == 8 and Wout == 8)


def gemm_bf16_splitk(x, w_packed, M, Cin, Cout, dst, splits, lda=None, accumulate=True):
    """dst[M][Cout] (+)= x @ W on the bf16 arm with the reduction split over ``splits`` slabs: the split-K of a weight-gradient GEMM
    (few output tiles, a reduction of tens of thousands of rows) expressed as a BATCHED launch of vf_gemm_bf16 — batch entry s
    reads columns [s Cin/splits, (s+1) Cin/splits) of x and the matching 64-deep chunk range of the chunk-major packing — followed by
    the fixed-order slab sum (deterministic).  Cin / splits must be a multiple of 64 (the packing's chunk)."""
    lib = _lib.load()
    ks = Cin // splits
    if splits < 1 or ks * splits != Cin or ks % 64:
        raise _lib.VfError(f'gemm_bf16_splitk: Cin = {Cin} does not split into {splits} ranges of whole 64-deep chunks')
    nb = (Cout + 127) // 128
    slabs = torch.empty((splits, M, Cout), dtype=torch.float32, device=x.device)
    igemm(x, w_packed, M, ks, Cout, slabs, lda=Cin if lda is None else lda, bf16=True, batch=splits, stride_x=ks,
          stride_w=(ks // 64) * nb * 64 * 128, stride_out=M * Cout)
    check(lib.vf_sum_slabs_f32(_p(slabs), splits, M * Cout, M * Cout, _p(_f32(dst)), 1 if accumulate else 0, _stream()),
          'vf_sum_slabs_f32')
    return dst


TN_TARGET_WORKGROUPS = 256             # row-range splits of the weight-gradient GEMM: tiles x splits ~ one workgroup per CU when the launch has the machine to
TN_TARGET_WORKGROUPS_BESIDE = 192      # itself; ~ three quarters of the CUs when it is issued beside another GEMM (the trainer's second stream: the dX GEMM of the
                                       # same layer runs on the main stream).  Every split writes an fp32 slab that vf_sum_slabs_f32 folds.  Round 6, in-process
                                       # alternation of the training step over the "beside" value: 256 (7 splits of the 36-tile layers) 19.72 ms, 224 (6) 19.71,
                                       # 208 / 192 (5) 19.47-19.49, 176 (4) 19.65, 128 19.85 — five slabs instead of seven are 29 % less slab traffic and the
                                       # parallelism they give up was not there to have (profiles/r6_small_kernels.txt).  Alone (the serialised step bench.py
                                       # instruments for the roofline section) the full-machine split is the faster one.


def gemm_tn_bf16_supported(x, M, K, N):
    return x.dtype == torch.bfloat16 and gemm_tn_bf16_shape_ok(M, K, N)


def gemm_tn_bf16_shape_ok(M, K, N):
    """shapes vf_gemm_tn_bf16 takes (csrc/gemm_tn_bf16.hip): 256-wide tiles of dW, 64-row chunks, 32-bit offsets into x"""
    return K % 256 == 0 and N % 256 == 0 and M % 64 == 0 and M * K * 2 < 2 ** 31


def gemm_g256_shape_ok(M, K, N):
    """shapes the 256-tile LDS-DMA bf16 GEMM takes (csrc/gemm_bf16_g256.hip: vf_gemm_bf16_g256_launch) — the only kernel behind the fused
    epilogues EPI_GELU_DUAL, EPI_GELU_BWD with a bf16 pre-activation, and the fused output dropout"""
    return N % 256 == 0 and K % 64 == 0 and M >= 256 and M * K * 2 < 2 ** 31 and K * N * 2 < 2 ** 31


def gemm_drop_supported(M, K, N, row0=0):
    """the fused output dropout of vf_gemm_bf16 (igemm(drop=...)): 256-tile shapes, 32-bit mask group indices"""
    return gemm_g256_shape_ok(M, K, N) and row0 % 4 == 0 and ((M + row0 + 3) // 4) * N < 2 ** 32


def gemm_tn_bf16(x16, dy, M, K, N, dw, db=None, accumulate=True, beside_another_gemm=False):
    """dw[K][N] (+)= x16^T @ dy and db[N] (+)= column sums of dy, straight from the row-major bf16 activation x16 [M][K] and the fp32
    gradient dy [M][N] (csrc/gemm_tn_bf16.hip): split-K slabs folded in slab order (deterministic)."""
    lib = _lib.load()
    _chk(x16, torch.bfloat16, 'x16')
    tiles = (K // 256) * (N // 256)
    target = TN_TARGET_WORKGROUPS_BESIDE if beside_another_gemm else TN_TARGET_WORKGROUPS
    splits = max(1, min(M // 64, target // tiles if tiles <= target else 1))
    # one array of `splits` records (weight slab | bias slab): where the gradient buffer holds the bias right behind the weight (the trainer's
    # flat buffer does), ONE slab sum folds both
    rec = K * N + (N if db is not None else 0)
    ws = torch.empty((splits, rec), dtype=torch.float32, device=dy.device)
    y16 = dy.dtype == torch.bfloat16
    check(lib.vf_gemm_tn_bf16(_p(x16), x16.stride(0), _p(dy if y16 else _f32(dy)), 1 if y16 else 0, dy.stride(0), M, K, N, splits, _p(ws),
                              ws.data_ptr() + K * N * 4 if db is not None else None, rec, _stream()), 'vf_gemm_tn_bf16')
    acc = 1 if accumulate else 0
    _f32(dw)
    if db is not None and _f32(db).data_ptr() == dw.data_ptr() + K * N * 4 and dw.is_contiguous():
        check(lib.vf_sum_slabs_f32(_p(ws), splits, rec, rec, _p(dw), acc, _stream()), 'vf_sum_slabs_f32')
    else:
        check(lib.vf_sum_slabs_f32(_p(ws), splits, rec, K * N, _p(dw), acc, _stream()), 'vf_sum_slabs_f32')
        if db is not None:
            check(lib.vf_sum_slabs_f32(ws.data_ptr() + K * N * 4, splits, rec, N, _p(db), acc, _stream()), 'vf_sum_slabs_f32')
    return dw


def igemm(x, w_packed, M, Cin, Cout, out, bias=None, res=None, mode=MODE_GEMM, epilogue=EPI_NONE,
          pro=None, pro_swish=False, pro_rows_per_img=0, Hin=0, Win=0, Hout=0, Wout=0,
          lda=None, ldc=None, ldr=None, batch=1, stride_x=0, stride_w=0, stride_out=0, stride_res=0, bf16=False, x6=False, gn_part=None, split_k=0,
          x3h=False, a16=False, o16=False, out_aux=None, res16=False, drop=None, gelu_grad=False):
    """``bf16=True``: w_packed is a bf16 packing (pack_*_bf16) and the launch goes to the bf16-MFMA arm
    (vf_gemm_bf16 / vf_conv3_halo_bf16); unsupported shapes raise (no silent fallback).
    ``x6=True``: w_packed is the 3-plane split packing (pack_conv3_x6) and the launch goes to the fp32-equivalent
    split-bf16 kernel (vf_conv3_halo_x6).
    ``gn_part``: fp32 [Nimg][halo_gn_slots(Hout, Wout)][32][2] buffer that receives the GroupNorm partial statistics of the
    output (halo kernels only; reduce with groupnorm_finalize).
    ``out_aux`` (with ``epilogue=EPI_GELU_DUAL``, bf16 arm, bf16 x, fp32 or bf16 out): bf16 [M][ldc] that receives gelu(out).
    ``res16`` (with ``epilogue=EPI_GELU_BWD``): the pre-activation behind ``res`` was saved as bf16.
    ``gelu_grad`` (round 6, 256-tile shapes): with ``EPI_GELU_DUAL`` and bf16 out, ``out`` receives gelu'(x @ W + b) instead of the pre-activation;
    with ``EPI_GELU_BWD`` and ``res16``, ``res`` holds that saved derivative and the epilogue multiplies by it instead of evaluating gelu' again.
    ``drop`` = (rate, seed, site), bf16 arm with bf16 x and fp32 out only: the training step's output dropout in the epilogue, before the
    residual joins (vf_igemm_args.drop_rate; the masks of train_ops.dropout_add with cols = Cout).  Shapes outside the 256-tile kernel
    raise 'unsupported' (callers run the GEMM, then dropout_add)."""
    lib = _lib.load()
    if not 0 <= M < 2 ** 31 or max(Cin, Cout, Hin, Win, Hout, Wout, batch) >= 2 ** 31:
        # vf_igemm_args carries 32-bit row / channel counts (byte offsets inside the kernels are 64-bit): refuse instead of wrapping
        raise _lib.VfError(f'igemm: M = {M} rows do not fit the int32 fields of vf_igemm_args — split the launch '
                           '(VQGAN(max_images_per_call=...))')
    a = VfIgemmArgs()
    a.x = x.data_ptr()
    a.w_packed = w_packed.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.res = res.data_ptr() if res is not None else None
    a.out = out.data_ptr()
    if out_aux is not None:
        _chk(out_aux, torch.bfloat16, 'out_aux')
        if out_aux.shape != out.shape or not out_aux.is_contiguous():
            raise _lib.VfError('out_aux must have the shape and row stride of out')
        a.out_aux = out_aux.data_ptr()
    if pro is not None:
        mean_c, scale_c, beta = pro
        a.pro_mean, a.pro_scale, a.pro_beta = mean_c.data_ptr(), scale_c.data_ptr(), beta.data_ptr()
    a.pro_swish = 1 if pro_swish else 0
    a.pro_rows_per_img = pro_rows_per_img
    a.mode, a.epilogue = mode, epilogue
    a.M, a.Cin, a.Cout = M, Cin, Cout
    a.Hin, a.Win, a.Hout, a.Wout = Hin, Win, Hout, Wout
    a.lda = Cin if lda is None else lda
    a.ldc = Cout if ldc is None else ldc
    a.ldr = (Cout if ldr is None else ldr)
    a.batch = batch
    a.stride_x, a.stride_w, a.stride_out, a.stride_res = stride_x, stride_w, stride_out, stride_res
    a.reserved0 = split_k                     # vf_gemm_x6: number of split-K slabs written at out + s*stride_out
    if gn_part is not None:
        a.gn_part = _f32(gn_part).data_ptr()
        a.gn_slots = gn_part.shape[1]
    conv16 = False
    if a16 or o16:                            # bf16 activations in / out: the bf16 arm only (dtype flags in reserved0)
        if not bf16 or mode not in (MODE_GEMM, MODE_CONV3_S1, MODE_CONV3_UP2):
            raise _lib.VfError('a16 / o16 need bf16=True and a GEMM or 3x3 stride-1 / upsample convolution')
        _chk(x, torch.bfloat16 if a16 else torch.float32, 'x')
        _chk(out, torch.bfloat16 if o16 else torch.float32, 'out')
        a.reserved0 = (1 if a16 else 0) | (2 if o16 else 0)
        conv16 = mode != MODE_GEMM
        if conv16:                            # vf_conv3_halo_bf16: bf16 out means a bf16 residual too (the same activation stream)
            if a16 and not o16:
                raise _lib.VfError('the bf16 convolution takes bf16 activations in only together with bf16 out')
            if res is not None:
                _chk(res, torch.bfloat16 if o16 else torch.float32, 'res')
    if drop is not None and drop[0]:
        if not (bf16 and a16 and not o16 and mode == MODE_GEMM and epilogue == EPI_NONE):
            raise _lib.VfError('drop needs the bf16 GEMM with bf16 x, fp32 out and no epilogue function')
        a.drop_rate, a.drop_seed, a.drop_site = float(drop[0]), int(drop[1]) & 0xFFFFFFFF, int(drop[2])
        a.drop_row0 = int(drop[3]) if len(drop) > 3 else 0          # first row's index in the global batch (data-parallel step)
    if res16:
        if not (a16 and o16 and epilogue == EPI_GELU_BWD and res is not None):
            raise _lib.VfError('res16 is the bf16 pre-activation of EPI_GELU_BWD (bf16 in, bf16 out)')
        _chk(res, torch.bfloat16, 'res')
        a.reserved0 |= 4
    if gelu_grad:
        if not ((epilogue == EPI_GELU_DUAL and a16 and o16) or (epilogue == EPI_GELU_BWD and res16)):
            raise _lib.VfError('gelu_grad: EPI_GELU_DUAL with bf16 in / out, or EPI_GELU_BWD with res16')
        a.reserved0 |= 8
    for t in ((None if a16 else x), (None if o16 else out), bias, (None if (res16 or conv16) else res)):
        if t is not None:
            _f32(t)
    if x3h:                                   # w_packed = pack_conv3_x3h / pack_dense_*_x3h: the 3-product split-fp16 kernels
        _chk(w_packed, torch.float16, 'w_packed')
        if mode == MODE_GEMM:
            check(lib.vf_gemm_x3h(ctypes.byref(a), _stream()), 'vf_gemm_x3h')
        else:
            check(lib.vf_conv3_halo_x3h(ctypes.byref(a), _stream()), 'vf_conv3_halo_x3h')
        return out
    if x6:
        _chk(w_packed, torch.bfloat16, 'w_packed')
        if mode == MODE_GEMM:
            check(lib.vf_gemm_x6(ctypes.byref(a), _stream()), 'vf_gemm_x6')
        else:
            check(lib.vf_conv3_halo_x6(ctypes.byref(a), _stream()), 'vf_conv3_halo_x6')
        return out
    if bf16:
        _chk(w_packed, torch.bfloat16, 'w_packed')
        if mode == MODE_GEMM:
            check(lib.vf_gemm_bf16(ctypes.byref(a), _stream()), 'vf_gemm_bf16')
        else:
            check(lib.vf_conv3_halo_bf16(ctypes.byref(a), _stream()), 'vf_conv3_halo_bf16')
        return out
    _f32(w_packed)
    check(lib.vf_igemm_f32(ctypes.byref(a), _stream()), 'vf_igemm_f32')
    return out


def conv3_small_cout_supported(mode, Cin, Cout, H, W):
    # the kernel keeps ALL weights in LDS (Cin / 32 x 4608 B, at most 96 KiB: Cin <= 672); wider inputs take the implicit-GEMM path
    return mode == MODE_CONV3_S1 and 1 <= Cout <= 4 and Cin % 32 == 0 and Cin // 32 * 4608 <= 96 * 1024 and H % 8 == 0 and W % 32 == 0


def conv3_small_cout(x, w_oihw, bias, n_img, H, W, Cin, Cout, pro=None, pro_swish=True, out=None):
    """3x3 conv to <= 4 channels (decoder conv_out) with the fused GroupNorm(+swish) prologue; x NHWC rows, fp32 or bf16"""
    lib = _lib.load()
    if out is None:
        out = torch.empty((n_img * H * W, Cout), dtype=torch.float32, device=x.device)
    pm, ps, pb = (None, None, None) if pro is None else pro
    x16 = x.dtype == torch.bfloat16
    check(lib.vf_conv3_small_cout_f32(_p(x if x16 else _f32(x)), _p(_f32(w_oihw)), _p(_f32(bias)) if bias is not None else None,
                                      _p(pm) if pm is not None else None, _p(ps) if ps is not None else None,
                                      _p(pb) if pb is not None else None, 1 if pro_swish else 0, _p(out), n_img, H, W, Cin, Cout,
                                      1 if x16 else 0, _stream()), 'vf_conv3_small_cout_f32')
    return out


def pack_conv_in_x3h(w_oihw):
    """encoder.conv_in weight [Cout][3][3][3] -> split-fp16 packing for the matrix-pipe form of conv_in (Cout % 128 == 0)"""
    lib = _lib.load()
    w = _f32(w_oihw).contiguous()
    out = torch.empty(int(lib.vf_conv_in_x3h_packed_elems(w.shape[0])), dtype=torch.float16, device=w.device)
    check(lib.vf_conv_in_x3h_pack(_p(w), _p(out), w.shape[0], _stream()), 'vf_conv_in_x3h_pack')
    return out


def conv_in_x3h_supported(H, W, Cout):
    return H % 8 == 0 and W % 16 == 0 and Cout % 128 == 0


def conv_in(img, w_oihw, bias, n_img, H, W, Cout, out=None, wp3h=None, gn_part=None):
    """img: uint8 NHWC [n,H,W,3] (TF evaluator entry) or float32 NHWC already in [-1,1].  ``wp3h`` (pack_conv_in_x3h): run on the
    matrix pipe (fp32-equivalent x3h arithmetic), optionally emitting the GroupNorm partial statistics of the output (gn_part)"""
    lib = _lib.load()
    if out is None:
        out = torch.empty((n_img, H, W, Cout), dtype=torch.float32, device=img.device)
    if img.dtype == torch.uint8:
        u8, f32 = _p(_chk(img, torch.uint8)), None
    else:
        u8, f32 = None, _p(_f32(img))
    if wp3h is not None:
        _chk(wp3h, torch.float16, 'wp3h')
        check(lib.vf_conv_in_x3h(u8, f32, _p(wp3h), _p(bias), _p(out), _p(gn_part) if gn_part is not None else None,
                                 gn_part.shape[1] if gn_part is not None else 0, n_img, H, W, Cout, _stream()), 'vf_conv_in_x3h')
        return out
    if gn_part is not None:
        raise _lib.VfError('conv_in: fused GroupNorm statistics need the x3h form (wp3h)')
    check(lib.vf_conv_in_u8_f32(u8, f32, _p(_f32(w_oihw)), _p(bias), _p(out), n_img, H, W, Cout, _stream()),
          'vf_conv_in_u8_f32')
    return out


# ------------------------------------------------------------------ GroupNorm
def groupnorm_stats(x, gamma, n_img, HW, C, groups=32, eps=1e-6):
    lib = _lib.load()
    mean_c = torch.empty((n_img, C), dtype=torch.float32, device=x.device)
    scale_c = torch.empty((n_img, C), dtype=torch.float32, device=x.device)
    ws = torch.empty(int(lib.vf_groupnorm_workspace_bytes(n_img, HW, C)), dtype=torch.uint8, device=x.device)
    check(lib.vf_groupnorm_stats_f32(_p(_f32(x)), _p(_f32(gamma)), n_img, HW, C, groups, eps, _p(mean_c), _p(scale_c),
                                     _p(ws), _stream()), 'vf_groupnorm_stats_f32')
    return mean_c, scale_c


def halo_gn_slots(Hout, Wout):
    """partial-statistics slots per image the halo conv kernels write for an Hout x Wout output (0 = not a halo shape)"""
    return int(_lib.load().vf_conv3_halo_gn_slots(Hout, Wout))


def new_gn_part(n_img, Hout, Wout, device):
    """buffer for igemm(..., gn_part=...): every slot is written by the producing conv, no initialisation needed"""
    return torch.empty((n_img, halo_gn_slots(Hout, Wout), 32, 2), dtype=torch.float32, device=device)


def groupnorm_finalize(part, gamma, n_img, HW, C, groups=32, eps=1e-6):
    """second half of groupnorm_stats from partials written by a conv's fused epilogue (the activation is not re-read)"""
    lib = _lib.load()
    mean_c = torch.empty((n_img, C), dtype=torch.float32, device=part.device)
    scale_c = torch.empty((n_img, C), dtype=torch.float32, device=part.device)
    check(lib.vf_groupnorm_finalize_f32(_p(_f32(part)), _p(_f32(gamma)), n_img, HW, C, groups, part.shape[1], eps,
                                        _p(mean_c), _p(scale_c), _stream()), 'vf_groupnorm_finalize_f32')
    return mean_c, scale_c


def groupnorm_apply(x, mean_c, scale_c, beta, n_img, HW, C, swish, out=None):
    lib = _lib.load()
    if out is None:
        out = torch.empty_like(x)
    check(lib.vf_groupnorm_apply_f32(_p(_f32(x)), _p(mean_c), _p(scale_c), _p(_f32(beta)), _p(out), n_img, HW, C,
                                     1 if swish else 0, _stream()), 'vf_groupnorm_apply_f32')
    return out


# ------------------------------------------------------------------ codebook
def vq_pack_codebook(E):
    lib = _lib.load()
    E = _f32(E).contiguous()
    D, Kc = E.shape
    out = torch.empty(int(lib.vf_vq_packed_floats(D, Kc)), dtype=torch.float32, device=E.device)
    check(lib.vf_vq_pack_codebook_f32(_p(E), _p(out), D, Kc, _stream()), 'vf_vq_pack_codebook_f32')
    e_sq = torch.empty(Kc, dtype=torch.float32, device=E.device)
    check(lib.vf_colsumsq_f32(_p(E), _p(e_sq), D, Kc, _stream()), 'vf_colsumsq_f32')
    return out, e_sq


def vq_argmin(z_rows, E_packed, e_sq, D, Kc):
    lib = _lib.load()
    z_rows = _f32(z_rows)
    M = z_rows.numel() // D
    idx = torch.empty(M, dtype=torch.int64, device=z_rows.device)
    check(lib.vf_vq_argmin_f32(_p(z_rows), _p(E_packed), _p(e_sq), M, D, Kc, _p(idx), _stream()), 'vf_vq_argmin_f32')
    return idx


def vq_filter_supported(D, Kc):
    """shape rules of the filtered lookup (vf_vq_argmin_filtered_f32): otherwise call vq_argmin (same result, f32 MFMA)"""
    return int(_lib.load().vf_vq_filter_packed_bytes(D, Kc)) > 0


def vq_filter_pack(E):
    """reference ``embeddings`` [D][Kc] -> the filtered lookup's blob (fp16 fragment tiles, fp32 transpose, e_sq, window constants)"""
    lib = _lib.load()
    E = _f32(E).contiguous()
    D, Kc = E.shape
    n = int(lib.vf_vq_filter_packed_bytes(D, Kc))
    if n == 0:
        raise _lib.VfError(f'vq_filter_pack: unsupported codebook shape D={D}, Kc={Kc} (need D == 256, Kc % 32 == 0, Kc <= 1024)')
    out = torch.empty(n, dtype=torch.uint8, device=E.device)
    check(lib.vf_vq_filter_pack(_p(E), _p(out), D, Kc, _stream()), 'vf_vq_filter_pack')
    return out


def vq_argmin_filtered(z_rows, blob, D, Kc, stats=None):
    """codebook lookup through the fp16 candidate filter + exact fp32 re-rank: the same indices as vq_argmin, bit for bit.
    ``stats``: optional zero-initialised int32[4] device tensor receiving (filter-certified rows, re-ranked rows, exact distance
    evaluations, fully scanned rows)"""
    lib = _lib.load()
    z_rows = _f32(z_rows)
    M = z_rows.numel() // D
    idx = torch.empty(M, dtype=torch.int64, device=z_rows.device)
    check(lib.vf_vq_argmin_filtered_f32(_p(z_rows), _p(_chk(blob, torch.uint8, 'blob')), M, D, Kc, _p(idx),
                                        _p(_chk(stats, torch.int32, 'stats')) if stats is not None else None, _stream()),
          'vf_vq_argmin_filtered_f32')
    return idx


def codebook_gather(E, idx, D, Kc):
    lib = _lib.load()
    idx = _chk(idx, torch.int64, 'codes').contiguous()
    M = idx.numel()
    out = torch.empty((M, D), dtype=torch.float32, device=idx.device)
    check(lib.vf_codebook_gather_f32(_p(_f32(E)), _p(idx), _p(out), M, D, Kc, _stream()), 'vf_codebook_gather_f32')
    return out


# ------------------------------------------------------------------ attention / transformer glue
def attn_blockcausal(q, k, v, out, B, H, T, L, ldq, ldk, ldv, ldo, scale=1.0, skip_masked=True, twin_view=-1, bf16=False, x6=False,
                     fp8=False):
    """``bf16`` / ``fp8``: the tolerance arms (bf16 or OCP e4m3 operands, fp32 softmax; csrc/attention_lp.hip, and for bf16 tensors with
    64-token views the LDS-DMA kernel of csrc/attention_dma.hip); ``x6``: fp32-equivalent; default: native f32 MFMA."""
    lib = _lib.load()
    if fp8:
        bf16 = True
    if x6:
        check(lib.vf_attn_blockcausal_x6(_p(_f32(q)), _p(_f32(k)), _p(_f32(v)), _p(_f32(out)), B, H, T, L, ldq, ldk, ldv,
                                         ldo, scale, 1 if skip_masked else 0, twin_view, _stream()),
              'vf_attn_blockcausal_x6')
        return out
    if bf16:
        o16 = out.dtype == torch.bfloat16          # bf16 output for a bf16-GEMM consumer (igemm(..., a16=True))
        i16 = q.dtype == torch.bfloat16            # bf16 q/k/v: the fused c_attn output written by its GEMM with o16=True
        for t in (q, k, v):
            _chk(t, torch.bfloat16 if i16 else torch.float32, 'q/k/v')
        fn = lib.vf_attn_blockcausal_fp8 if fp8 else lib.vf_attn_blockcausal_bf16_v2
        check(fn(_p(q), _p(k), _p(v), 1 if i16 else 0, _p(out if o16 else _f32(out)), 1 if o16 else 0, B, H, T, L,
                 ldq, ldk, ldv, ldo, scale, 1 if skip_masked else 0, twin_view, _stream()), 'vf_attn_blockcausal_lp')
        return out
    check(lib.vf_attn_blockcausal_f32(_p(_f32(q)), _p(_f32(k)), _p(_f32(v)), _p(_f32(out)), B, H, T, L, ldq, ldk, ldv,
                                      ldo, scale, 1 if skip_masked else 0, twin_view, _stream()),
          'vf_attn_blockcausal_f32')
    return out


def attn_spatial_supported(HW, C):
    return (HW, C) in ((256, 256), (64, 512), (64, 256))


def attn_spatial(qkv, n_img, HW, C, scale, out=None, x3h=False):
    """fused single-head attention of the VQGAN AttnBlock: qkv [n_img*HW][3C] (q|k|v) -> [n_img*HW][C]; scores stay on chip.
    ``x3h``: the fp32-equivalent three-product fp16 form (vf_attn_spatial_x3h) instead of the native f32 MFMA"""
    if out is None:
        out = torch.empty((n_img * HW, C), dtype=torch.float32, device=qkv.device)
    fn = _lib.load().vf_attn_spatial_x3h if x3h else _lib.load().vf_attn_spatial_f32
    check(fn(_p(_f32(qkv)), _p(_f32(out)), n_img, HW, C, qkv.stride(0), out.stride(0), scale, _stream()),
          'vf_attn_spatial_x3h' if x3h else 'vf_attn_spatial_f32')
    return out


def softmax_rows_(x, rows, n, scale=1.0):
    check(_lib.load().vf_softmax_rows_f32(_p(_f32(x)), rows, n, scale, _stream()), 'vf_softmax_rows_f32')
    return x


def layernorm(x, gamma, beta, rows, d, eps=1e-5, out=None, out_bf16=False):
    """``out_bf16``: write the normalised rows as bf16 (for a bf16-GEMM consumer: ``igemm(..., bf16=True, a16=True)``)"""
    if out_bf16:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if out is None else _chk(out, torch.bfloat16, 'out')
        check(_lib.load().vf_layernorm_bf16out_f32(_p(_f32(x)), _p(_f32(gamma)), _p(_f32(beta)), _p(out), rows, d, eps, _stream()),
              'vf_layernorm_bf16out_f32')
        return out
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().vf_layernorm_f32(_p(_f32(x)), _p(_f32(gamma)), _p(_f32(beta)), _p(out), rows, d, eps, _stream()),
          'vf_layernorm_f32')
    return out


def embed_sum(ids_i32, wte, wpe, add, BS, L, d, vocab):
    out = torch.empty((BS * L, d), dtype=torch.float32, device=wte.device)
    check(_lib.load().vf_embed_sum_f32(_p(_chk(ids_i32, torch.int32, 'ids')), _p(_f32(wte)), _p(_f32(wpe)),
                                       _p(_f32(add)), _p(out), BS, L, d, vocab, _stream()), 'vf_embed_sum_f32')
    return out


def dense_small_k(x, W, b, rows, K, N, gelu):
    out = torch.empty((rows, N), dtype=torch.float32, device=x.device)
    check(_lib.load().vf_dense_small_k_gelu_f32(_p(_f32(x)), _p(_f32(W)), _p(b), _p(out), rows, K, N,
                                                1 if gelu else 0, _stream()), 'vf_dense_small_k_gelu_f32')
    return out


def argmax_rows(x, rows, n, ld=None):
    idx = torch.empty(rows, dtype=torch.int64, device=x.device)
    check(_lib.load().vf_argmax_rows_f32(_p(_f32(x)), rows, n, n if ld is None else ld, _p(idx), _stream()),
          'vf_argmax_rows_f32')
    return idx


def lmhead_argmax_supported(K, N):
    return K in (128, 768) and N % 128 == 0


def lmhead_argmax_bf16(h, w_packed_bf16, M, K, N, want_max=False):
    """fused tied LM head + arg-max (bf16 arm): h [M][K] fp32 or bf16 rows, w_packed_bf16 = pack_dense_nk_bf16(wte, n_rows=N).
    Returns idx int64 [M] (and the winning logits if ``want_max``); the [M][N] logits are never materialised."""
    idx = torch.empty(M, dtype=torch.int64, device=h.device)
    mx = torch.empty(M, dtype=torch.float32, device=h.device) if want_max else None
    h16 = h.dtype == torch.bfloat16
    _chk(h, torch.bfloat16 if h16 else torch.float32, 'h')
    check(_lib.load().vf_lmhead_argmax_bf16(_p(h), 1 if h16 else 0, h.stride(0), _p(_chk(w_packed_bf16, torch.bfloat16, 'w_packed')), M, K, N,
                                            _p(idx), _p(mx) if mx is not None else None, _stream()), 'vf_lmhead_argmax_bf16')
    return (idx, mx) if want_max else idx


def postprocess_u8(x):
    x = _f32(x).contiguous()
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    check(_lib.load().vf_postprocess_u8(_p(x), _p(out), x.numel(), _stream()), 'vf_postprocess_u8')
    return out


def resize_u8(images, image_size, method=None):
    """``resize`` of viewformer/data/_common.py:47-61 on NHWC uint8 images [n,H,W,C]: identity when H == image_size, otherwise torch
    'nearest' when enlarging / bilinear (align_corners=False) when shrinking unless ``method`` says otherwise; uint8 out"""
    if method not in (None, 'nearest', 'bilinear'):
        raise ValueError("method must be None, 'nearest' or 'bilinear'")
    n, H, W, C = images.shape
    # resize() tests the NHWC batch's shape[-2] (= W), then resize_th() tests the NCHW tensor's shape[-2] (= H): either match
    # returns the frames untouched (data/_common.py:26-27,54-55); the enlarge/shrink choice below uses H (:34-37)
    if W == image_size or H == image_size:
        return images
    if method is None:
        method = 'nearest' if image_size > H else 'bilinear'
    src = _chk(images.contiguous(), torch.uint8, 'images')
    out = torch.empty((n, image_size, image_size, C), dtype=torch.uint8, device=images.device)
    check(_lib.load().vf_resize_u8(_p(src), _p(out), n, H, W, image_size, image_size, C, 1 if method == 'bilinear' else 0, _stream()),
          'vf_resize_u8')
    return out
