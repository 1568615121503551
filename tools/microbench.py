#!/usr/bin/env python
"""Per-kernel microbenchmark (GPU): times the hot kernels at the bench's dominant shapes with HIP
events.  Used under rocprofv3 for the profiles/ summaries.  python tools/microbench.py [names...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from viewformer_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=8, warm=2):
    # the first launches of a process run at ramping clocks (measured: ~10 % slow): warm for >= 0.3 s of GPU work
    t0 = time.time()
    n = 0
    while n < warm or time.time() - t0 < 0.3:
        fn()
        torch.cuda.synchronize()
        n += 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def conv_s2(n_img=224, C=128, H=128, x6=True, x3h=False):
    x = torch.randn(n_img * H * H, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.03
    x6 = x6 and not x3h
    wp = ops.pack_conv3_x3h(w) if x3h else ops.pack_conv3_x6(w) if x6 else ops.pack_conv_oihw(w)
    b = torch.randn(C, device=dev)
    Ho = H // 2
    M = n_img * Ho * Ho
    out = torch.empty(M, C, device=dev)
    ms = timeit(lambda: ops.igemm(x, wp, M, C, C, out, bias=b, mode=ops.MODE_CONV3_S2PAD, Hin=H, Win=H, Hout=Ho, Wout=Ho, x6=x6, x3h=x3h))
    print(f'conv3x3 s2{" x3h" if x3h else " x6" if x6 else ""} {C}->{C} @{H}^2->{Ho}^2 x{n_img}: {ms:.3f} ms  {2.0 * M * C * C * 9 / ms / 1e9:.1f} TF')


def conv(n_img=56, C=128, H=128, pro=True, x6=False, bf16=False, x3h=False, io16=False):
    x = torch.randn(n_img * H * H, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.03
    if os.environ.get('VF_MB_ZERO') == '1':           # zero operands: no datapath toggling -> how much of the time is power (DVFS)?
        x.zero_(); w.zero_(); w[0, 0, 0, 0] = 1.0
    wp = ops.pack_conv3_x3h(w) if x3h else ops.pack_conv3_x6(w) if x6 else ops.pack_conv3_bf16(w) if bf16 else ops.pack_conv_oihw(w)
    b = torch.randn(C, device=dev)
    out = torch.empty_like(x)
    prol = None
    if pro:
        g = torch.ones(C, device=dev)
        m, s = ops.groupnorm_stats(x, g, n_img, H * H, C)
        prol = (m, s, torch.zeros(C, device=dev))
    M = n_img * H * H
    if io16:                                          # bf16 activations in / out (the decoder's act16 stream)
        x = x.to(torch.bfloat16)
        out = torch.empty_like(x)
    part = ops.new_gn_part(n_img, H, H, dev) if os.environ.get('VF_MB_GN') == '1' else None      # fused GroupNorm partials of the output (as in the model)
    res = None if os.environ.get('VF_MB_NORES') == '1' else x
    ms = timeit(lambda: ops.igemm(x, wp, M, C, C, out, bias=b, res=res, mode=ops.MODE_CONV3_S1, pro=prol, pro_swish=True,
                                  Hin=H, Win=H, Hout=H, Wout=H, x6=x6, bf16=bf16, x3h=x3h, a16=io16, o16=io16, gn_part=part))
    fl = 2.0 * M * C * C * 9
    print(f'conv3x3{" x3h" if x3h else " x6" if x6 else " bf16" if bf16 else ""}{" io16" if io16 else ""} {C}->{C} @{H}^2 x{n_img} pro={pro}: {ms:.3f} ms  {fl / ms / 1e9:.1f} TF')


def gemm(M=7168, K=768, N=3072, epi=0, arith='f32'):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(K, N, device=dev) * 0.02
    wp = {'f32': ops.pack_dense_kn, 'x6': ops.pack_dense_kn_x6, 'bf16': ops.pack_dense_kn_bf16, 'x3h': ops.pack_dense_kn_x3h}[arith](w)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    ms = timeit(lambda: ops.igemm(x, wp, M, K, N, out, bias=b, epilogue=epi, x6=arith == 'x6', bf16=arith == 'bf16', x3h=arith == 'x3h'))
    print(f'gemm[{arith}] {M}x{K}x{N} epi={epi}: {ms:.3f} ms  {2.0 * M * K * N / ms / 1e9:.1f} TF')


def gemm_1x1(M=229376, K=256, N=768, HW=256, pro=True):
    """the encoder's 1x1 convolutions on the x3h GEMM: the fused q|k|v projection of an AttnBlock (GroupNorm prologue) at the bench's size"""
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    wp = ops.pack_dense_nk_x3h(w)
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    prol = None
    if pro:
        n_img = M // HW
        m, s = ops.groupnorm_stats(x, torch.ones(K, device=dev), n_img, HW, K)
        prol = (m, s, torch.zeros(K, device=dev))
    ms = timeit(lambda: ops.igemm(x, wp, M, K, N, out, bias=b, pro=prol, pro_rows_per_img=HW if pro else 0, x3h=True), iters=20)
    by = M * K * 4 + M * N * 4
    print(f'gemm_x3h 1x1 {M}x{K}x{N} pro={pro}: {ms * 1e3:.1f} us  {2.0 * M * K * N / ms / 1e9:.1f} TF  {by / ms / 1e6:.0f} GB/s')


def attnsp(n=896, HW=256, C=256):
    """the encoder's AttnBlock core at the bench's size, native f32 MFMA vs x3h"""
    qkv = torch.randn(n * HW, 3 * C, device=dev) * 0.7
    for x3h in (False, True):
        ms = timeit(lambda: ops.attn_spatial(qkv, n, HW, C, C ** -0.5, x3h=x3h), iters=20)
        print(f'attn_spatial{" x3h" if x3h else " f32"} {n}x{HW}x{C}: {ms * 1e3:.1f} us  {4.0 * n * HW * HW * C / ms / 1e9:.1f} TF')


def convout(n_img=128, C=128, H=128):
    """the decoder's conv_out (128 -> 3, norm_out + swish fused) at the bench's size"""
    x = torch.randn(n_img * H * H, C, device=dev)
    w = torch.randn(3, C, 3, 3, device=dev) * 0.05
    b = torch.randn(3, device=dev)
    m, s = ops.groupnorm_stats(x, torch.ones(C, device=dev), n_img, H * H, C)
    pro = (m, s, torch.zeros(C, device=dev))
    ms = timeit(lambda: ops.conv3_small_cout(x, w, b, n_img, H, H, C, 3, pro=pro, pro_swish=True), iters=20)
    print(f'conv_out {C}->3 @{H}^2 x{n_img}: {ms * 1e3:.1f} us  {x.numel() * 4 / ms / 1e6:.0f} GB/s read')


def x3h_stamps(n_img=56, C=128, H=128):
    """stage-loop time vs chunk-barrier wait of the x3h 3x3 convolution (library built with -DVF_X3H_STAMPS), cycles per chunk per wave"""
    import numpy as np
    x = torch.randn(n_img * H * H, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.03
    wp = ops.pack_conv3_x3h(w)
    b = torch.randn(C, device=dev)
    out = torch.empty_like(x)
    g = torch.ones(C, device=dev)
    m, s = ops.groupnorm_stats(x, g, n_img, H * H, C)
    slots = ops.halo_gn_slots(H, H) if hasattr(ops, 'halo_gn_slots') else (H // 8) * (H // 16) * 2
    nwg = n_img * (H // 8) * (H // 16) * (C // 128)
    extra = (nwg * 16 + slots * 64 - 1) // (slots * 64)
    part = torch.zeros(n_img + extra, slots, 32, 2, device=dev)
    M = n_img * H * H
    for _ in range(3):
        ops.igemm(x, wp, M, C, C, out, bias=b, res=x, mode=ops.MODE_CONV3_S1, pro=(m, s, torch.zeros(C, device=dev)), pro_swish=True,
                  Hin=H, Win=H, Hout=H, Wout=H, x3h=True, gn_part=part)
    torch.cuda.synchronize()
    t = part.view(-1)[n_img * slots * 64:].view(torch.int32)[:nwg * 16].cpu().numpy().view(np.uint32).reshape(nwg, 4, 4).astype(np.float64)
    nch = C // 32
    print(f'x3h conv {C}->{C} @{H}^2 x{n_img}: {nwg} workgroups, {nch} chunks of 18 stages (216 MFMAs = 6912 pipe cycles per wave per chunk)')
    print('  stage loop   per chunk: mean %7.0f  p10 %7.0f  p90 %7.0f' % (t[:, :, 0].mean() / nch, np.percentile(t[:, :, 0], 10) / nch, np.percentile(t[:, :, 0], 90) / nch))
    print('  barrier wait per chunk: mean %7.0f  p10 %7.0f  p90 %7.0f' % (t[:, :, 1].mean() / nch, np.percentile(t[:, :, 1], 10) / nch, np.percentile(t[:, :, 1], 90) / nch))
    print('  tile head (first patch: load, transform, park, barrier): mean %7.0f  p10 %7.0f  p90 %7.0f' % (t[:, :, 2].mean(), np.percentile(t[:, :, 2], 10), np.percentile(t[:, :, 2], 90)))
    print('  epilogue (residual, stores accepted):                    mean %7.0f  p10 %7.0f  p90 %7.0f' % (t[:, :, 3].mean(), np.percentile(t[:, :, 3], 10), np.percentile(t[:, :, 3], 90)))
    tot = t[:, :, 0] + t[:, :, 1] + t[:, :, 2] + t[:, :, 3]
    print('  share of a wave\'s tile time: loop %.3f  barrier %.3f  head %.3f  epilogue %.3f  (tile %.0f cycles)' % (
        t[:, :, 0].sum() / tot.sum(), t[:, :, 1].sum() / tot.sum(), t[:, :, 2].sum() / tot.sum(), t[:, :, 3].sum() / tot.sum(), tot.mean()))


def gemm_tf(M=65536, only=None):
    """the four dense layers of one transformer block at the bench's size (128 scenes x 8 views x 64 tokens), bf16 arm with bf16
    activations: c_attn (fp32 or bf16 qkv out), attn.c_proj (+ residual), mlp.c_fc (GELU, bf16 out), mlp.c_proj (+ residual)"""
    d = 768
    for name, K, N, epi, o16, res in (('c_attn', d, 3 * d, 0, False, False), ('c_attn/o16', d, 3 * d, 0, True, False),
                                      ('attn.c_proj', d, d, 0, False, True), ('mlp.c_fc', d, 4 * d, 1, True, False),
                                      ('mlp.c_proj', 4 * d, d, 0, False, True)):
        if only and name != only:
            continue
        x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        w = torch.randn(K, N, device=dev) * 0.02
        if os.environ.get('VF_MB_ZERO') == '1':           # zero operands: how much of the time is the package power limit?
            x.zero_(); w.zero_()
        wp = ops.pack_dense_kn_bf16(w)
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if o16 else torch.float32)
        r = torch.randn(M, N, device=dev) if res else None
        ms = timeit(lambda: ops.igemm(x, wp, M, K, N, out, bias=b, epilogue=epi, res=r, bf16=True, a16=True, o16=o16), iters=20)
        by = M * K * 2 + K * N * 2 + M * N * (2 if o16 else 4) * (2 if res else 1)
        print(f'gemm_bf16 {name:12s} {M}x{K}x{N}: {ms * 1e3:7.1f} us  {2.0 * M * K * N / ms / 1e9:6.1f} TF = {2.0 * M * K * N / ms / 1e9 / 25:.1f} % of 2500;'
              f'  {by / 1e6:.0f} MB -> {by / ms / 1e6:.0f} GB/s')


def g256_stamps(M=65536, K=3072, N=768):
    """phase timeline of the 256-tile bf16 GEMM (library built with -DG256_STAMPS): per wave, cycles summed over the stages"""
    import ctypes
    import numpy as np
    from viewformer_amd import _lib
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    wp = ops.pack_dense_kn_bf16(torch.randn(K, N, device=dev) * 0.02)
    out = torch.empty(M, N, device=dev)
    nwg = (M // 256) * (N // 256)
    st = torch.zeros(nwg * 8 * 8, dtype=torch.int32, device=dev)
    a = ops.VfIgemmArgs()
    a.x, a.w_packed, a.out = x.data_ptr(), wp.data_ptr(), out.data_ptr()
    a.pro_beta = st.data_ptr()
    a.mode, a.epilogue, a.M, a.Cin, a.Cout, a.lda, a.ldc, a.ldr, a.batch, a.reserved0 = ops.MODE_GEMM, 0, M, K, N, K, N, N, 1, 1
    for _ in range(3):
        _lib.check(_lib.load().vf_gemm_bf16(ctypes.byref(a), ops._stream()), 'vf_gemm_bf16')
    torch.cuda.synchronize()
    t = st.cpu().numpy().view(np.uint32).reshape(nwg, 8, 8).astype(np.float64)
    ns = K // 64
    print(f'g256 {M}x{K}x{N}: {nwg} workgroups, {ns} stages; cycles per stage per wave (mean over waves; [wave_m 0 | wave_m 1])')
    for i, name in enumerate(('dma issue', 'first frags', 'mfma phase', 'dma wait', 'barrier wait')):
        v = t[:, :, i] / ns
        print('  %-12s mean %7.0f   [%7.0f | %7.0f]   p10 %7.0f p90 %7.0f' % (name, v.mean(), v[:, :4].mean(), v[:, 4:].mean(), np.percentile(v, 10), np.percentile(v, 90)))
    print('  main loop total per wave: mean %.0f cycles = %.0f per stage' % (t[:, :, 5].mean(), t[:, :, 5].mean() / ns))
    print('  epilogue per wave: issue mean %.0f cycles (p10 %.0f p90 %.0f), store drain mean %.0f (p10 %.0f p90 %.0f)' % (
        t[:, :, 6].mean(), np.percentile(t[:, :, 6], 10), np.percentile(t[:, :, 6], 90), t[:, :, 7].mean(), np.percentile(t[:, :, 7], 10),
        np.percentile(t[:, :, 7], 90)))
    ms = timeit(lambda: _lib.check(_lib.load().vf_gemm_bf16(ctypes.byref(a), ops._stream()), 'vf_gemm_bf16'))
    rounds = -(-nwg // 256)
    print('  wall %.1f us = %.1f us per round of 256 tiles' % (ms * 1e3, ms * 1e3 / rounds))


def vq(M=64 * 448):
    z = torch.randn(M, 256, device=dev) * 0.2
    E = torch.randn(256, 1024, device=dev) * 0.05
    Ep, esq = ops.vq_pack_codebook(E)
    ms = timeit(lambda: ops.vq_argmin(z, Ep, esq, 256, 1024))
    by = M * 256 * 4 + 256 * 1024 * 4 + M * 8
    print(f'vq_argmin M={M}: {ms:.4f} ms  {2.0 * M * 256 * 1024 / ms / 1e9:.1f} TF  {by / ms / 1e6:.1f} GB/s algorithmic')


def vqf(M=64 * 896, zscale=0.18):
    """filtered lookup at the bench's launch size (896 images); z at the encoder's magnitude (|z| ~ 3)"""
    z = torch.randn(M, 256, device=dev) * zscale
    E = (torch.rand(256, 1024, device=dev) * 2 - 1) * (3 ** 0.5) * 0.05
    blob = ops.vq_filter_pack(E)
    st = torch.zeros(4, dtype=torch.int32, device=dev)
    ops.vq_argmin_filtered(z, blob, 256, 1024, stats=st)
    ms = timeit(lambda: ops.vq_argmin_filtered(z, blob, 256, 1024), iters=20)
    by = M * 256 * 4 + 256 * 1024 * 4 + M * 8
    s = st.cpu().tolist()
    print(f'vq_filtered M={M}: {ms * 1e3:.1f} us  {2.0 * M * 256 * 1024 / ms / 1e9:.1f} TF (fp16 filter)  {by / ms / 1e6:.1f} GB/s algorithmic = '
          f'{by / ms / 1e6 / 8000 * 100:.1f} % of 8 TB/s; certified {s[0]} reranked {s[1]} exact evals {s[2]} scanned {s[3]}')


def vqf_stamps(M=64 * 896, zscale=0.18):
    """phase timeline of the filtered lookup (library built with -DVQF_STAMPS): s_memtime at phase boundaries, wave 0 of every workgroup"""
    import numpy as np
    z = torch.randn(M, 256, device=dev) * zscale
    E = (torch.rand(256, 1024, device=dev) * 2 - 1) * (3 ** 0.5) * 0.05
    blob = ops.vq_filter_pack(E)
    st = torch.zeros(4 + 1024 * 20, dtype=torch.int32, device=dev)
    for _ in range(3):
        st.zero_()
        ops.vq_argmin_filtered(z, blob, 256, 1024, stats=st)
    torch.cuda.synchronize()
    t = st[4:].cpu().numpy().view(np.uint64).reshape(1024, 10)[:min(1024, (M + 127) // 128)].astype(np.float64)
    # order of events: 0 start, 1 A ready, 2 main loop done, 6 barrier passed, 7 row max done, 8 window flags done, 3 pairs queued,
    # 9 first pass staged, 4 re-rank done, 5 end
    order = [0, 1, 2, 6, 7, 8, 3, 9, 4, 5]
    names = ['loadA', 'main', 'barrier', 'rowmax', 'flags', 'queue', 'stage1', 'rerank', 'final']
    print(f'M={M}: {len(t)} workgroups')
    for name, a, b in zip(names, order[:-1], order[1:]):
        ok = (t[:, a] > 0) & (t[:, b] > 0)
        col = (t[ok, b] - t[ok, a])
        if len(col):
            print('  %-8s n %4d  min %7.0f  median %7.0f  p90 %7.0f  max %7.0f' % (name, len(col), col.min(), np.median(col), np.percentile(col, 90), col.max()))
    print('  per-WG total: median %.0f max %.0f' % (np.median(t[:, 5] - t[:, 0]), (t[:, 5] - t[:, 0]).max()))


def attn(B=128, H=12, S=8, L=64, bf16=False, x6=False, fp8=False, twin=6, a16=False):
    """the bench's attention call: 128 scenes x 12 heads, the fused twin pass of 6 context views + MASK view + LOC view (T = 512);
    useful FLOPs = the (query view, key view) tile pairs the mask keeps"""
    d, T = H * 64, S * L
    qkv = torch.randn(B * T, 3 * d, device=dev) * 0.3
    if a16:
        qkv = qkv.to(torch.bfloat16)
    out = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16 if a16 else torch.float32)
    if a16:
        from viewformer_amd import _lib as _vl
        arm_dma = bool(_vl.load().vf_selected(_vl.SEL_ATTN_DMA))      # (VF_ATTN_DMA=0 in the environment is applied once, at load)
    ms = timeit(lambda: ops.attn_blockcausal(qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d], out, B, H, T, L, 3 * d, 3 * d,
                                             3 * d, d, 1.0, True, twin, bf16=bf16, x6=x6, fp8=fp8), iters=20)
    if twin <= -2:
        sv = -twin
        pairs = sv * (sv + 1) // 2 + (S // sv - 1) * sv * (sv + 1) // 2        # streams: branch position i sees i main views + itself
    else:
        pairs = S * (S + 1) // 2 if twin < 0 else (twin * (twin + 1) // 2 + (S - twin) * (twin + 1))
    useful = 4.0 * H * 64 * L * L * pairs * B
    arm = 'fp8' if fp8 else 'bf16' if bf16 else 'x6' if x6 else 'f32'
    peak = 2500.0 if (bf16 or fp8) else 2500.0 / 6 if x6 else 157.3
    if a16 and arm_dma:
        arm += ', q32' if _vl.load().vf_selected(_vl.SEL_ATTN_Q32) else ', q64'
    print(f'attn[{arm}{(",bf16 io, dma ring" if arm_dma else ",bf16 io") if a16 else ""}] B={B} H={H} T={T} twin={twin}: {ms * 1e3:.1f} us  {useful / ms / 1e9:.1f} TF useful = '
          f'{useful / ms / 1e9 / peak * 100:.1f} % of {peak:.0f}')


def attn_stamps(B=128, H=12, S=8, twin=6):
    """phase timeline of the LDS-DMA attention (library built with -DADMA_STAMPS: the log-sum-exp pointer carries the stamp buffer): per
    wave, cycles summed over its tile steps — wait for the tile's DMA, wait at the barrier, S MFMAs + softmax, V^T reads + P.V MFMAs"""
    import ctypes
    import numpy as np
    from viewformer_amd import _lib
    L, d, T = 64, H * 64, S * 64
    qkv = (torch.randn(B * T, 3 * d, device=dev) * 0.3).to(torch.bfloat16)
    out = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
    nq = (T + 255) // 256
    for q32 in (1, 0):
        _lib.select(_lib.SEL_ATTN_Q32, q32)
        nw = 8 if q32 else 4
        st = torch.zeros(B * H * nq * nw * 8, dtype=torch.int32, device=dev)
        P = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
        for _ in range(3):
            st.zero_()
            _lib.check(_lib.load().vf_attn_blockcausal_bf16_lse(P(qkv[:, d:2 * d]), P(qkv[:, 2 * d:]), P(qkv[:, :d]), P(out), P(st), B, H, T, L, 3 * d, 3 * d,
                                                               3 * d, d, 1.0, twin, 0.0, 0, 0, 0, ops._stream()), 'lse')
        torch.cuda.synchronize()
        t = st.cpu().numpy().view(np.uint32).reshape(nq, B, H, nw, 8).astype(np.float64)
        print(f'attention stamps B={B} H={H} T={T} twin={twin}, {"8 waves x 32 queries" if q32 else "4 waves x 64 queries"}: cycles per wave')
        for z in range(nq):
            tz = t[z].reshape(-1, nw, 8)
            vis = np.maximum(tz[:, :, 5], 1)
            print(f'  query block {z}: prologue (entry -> Q in registers) mean {tz[:, :, 6].mean():.0f}; tile loop mean {tz[:, :, 7].mean():.0f} '
                  f'(max wave {tz[:, :, 7].max():.0f}); visible tile steps per wave {tz[:, :, 5].mean(0).round(1).tolist()}')
            for i, name in enumerate(('dma wait', 'barrier wait', 'dma issue', 'S + softmax', 'V reads + PV')):
                per_wave = tz[:, :, i].mean(0)
                print('    %-13s per wave (summed over its tile steps): %s' % (name, ' '.join('%6.0f' % v for v in per_wave)))
            print('    S + softmax + PV per VISIBLE tile step: %s' % ' '.join('%6.0f' % v for v in ((tz[:, :, 3] + tz[:, :, 4]) / vis).mean(0)))
    _lib.select(_lib.SEL_ATTN_Q32, 1)


def gn(n_img=56, C=128, HW=16384):
    x = torch.randn(n_img * HW, C, device=dev)
    g = torch.ones(C, device=dev)
    ms = timeit(lambda: ops.groupnorm_stats(x, g, n_img, HW, C))
    print(f'gn_stats {n_img}x{HW}x{C}: {ms:.4f} ms  {x.numel() * 4 / ms / 1e6:.1f} GB/s')


def convin(n_img=224, H=128, C=128, x3h=False):
    img = torch.randint(0, 256, (n_img, H, H, 3), dtype=torch.uint8, device=dev)
    w = torch.randn(C, 3, 3, 3, device=dev) * 0.2
    b = torch.randn(C, device=dev)
    out = torch.empty((n_img, H, H, C), device=dev)
    wp = ops.pack_conv_in_x3h(w) if x3h else None
    ms = timeit(lambda: ops.conv_in(img, w, b, n_img, H, H, C, out=out, wp3h=wp))
    print(f'conv_in{" x3h" if x3h else ""} u8 {n_img}x{H}^2 -> {C}ch: {ms:.3f} ms  {out.numel() * 4 / ms / 1e6:.0f} GB/s written')


def clockprobe(n_img=56, C=128, H=128):
    """needs a -DVF_X6_CLOCKPROBE build (tools/variants.sh): shader clock during the x6 conv from one workgroup's
    s_memtime / s_memrealtime (100 MHz) stamps, delivered through the gn_part pointer"""
    x = torch.randn(n_img * H * H, C, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.03
    wp = ops.pack_conv3_x6(w)
    out = torch.empty_like(x)
    part = ops.new_gn_part(n_img, H, H, dev)
    g = torch.ones(C, device=dev)
    m, s = ops.groupnorm_stats(x, g, n_img, H * H, C)
    M = n_img * H * H
    fn = lambda: ops.igemm(x, wp, M, C, C, out, res=x, mode=ops.MODE_CONV3_S1, pro=(m, s, torch.zeros(C, device=dev)), pro_swish=True,
                           Hin=H, Win=H, Hout=H, Wout=H, x6=True, gn_part=part)
    ms = timeit(fn)
    torch.cuda.synchronize()
    d = part.view(-1)[:4].view(torch.int64).cpu().tolist()
    print(f'x6 conv {ms:.3f} ms; one workgroup: {d[0]} shader cycles in {d[1] / 100.0:.1f} us -> sclk ~ {d[0] / (d[1] / 100.0) / 1e3:.2f} GHz')


ALL = dict(clockprobe=clockprobe,
           convbf16=lambda: conv(32, 128, 128, bf16=True), convbf16_64=lambda: conv(32, 128, 64, bf16=True),
           convbf16_256=lambda: conv(32, 256, 32, bf16=True), convbf16_dec=lambda: conv(128, 128, 128, bf16=True), convbf16_dec_nopro=lambda: conv(128, 128, 128, bf16=True, pro=False), convbf16_dec512=lambda: conv(128, 512, 16, bf16=True),
           convbf16_io16=lambda: conv(128, 128, 128, bf16=True, io16=True), convbf16_io16_nopro=lambda: conv(128, 128, 128, bf16=True, io16=True, pro=False),
           convbf16_io16_64=lambda: conv(128, 128, 64, bf16=True, io16=True), convbf16_io16_256=lambda: conv(128, 256, 32, bf16=True, io16=True),
           attnbf16=lambda: attn(bf16=True), attnx6=lambda: attn(x6=True), attnfp8=lambda: attn(fp8=True), attnbf16_io16=lambda: attn(bf16=True, a16=True), attn_stamps=attn_stamps, attn_stamps_s20=lambda: attn_stamps(12, 12, 21, 19),
           attnbf16_train=lambda: attn(B=10, H=12, S=30, bf16=True, a16=True, twin=-10),
           attnbf16_s20=lambda: attn(B=12, S=21, twin=19, bf16=True), attnfp8_s20=lambda: attn(B=12, S=21, twin=19, fp8=True),
           convs2=lambda: conv_s2(x6=False), convs2x6=conv_s2, convs2x3h=lambda: conv_s2(x3h=True), convs2x3h_256=lambda: conv_s2(224, 256, 32, x3h=True), convs2x6_256=lambda: conv_s2(224, 256, 32),
           gemmx3h=lambda: gemm(16384, 768, 2304, arith='x3h'), gemmx3h_gelu=lambda: gemm(16384, 768, 3072, 1, 'x3h'),
           gemmx3h_k3072=lambda: gemm(16384, 3072, 768, arith='x3h'),
           gemmx6=lambda: gemm(16384, 768, 2304, arith='x6'), gemmx6_gelu=lambda: gemm(16384, 768, 3072, 1, 'x6'),
           gemmx6_k3072=lambda: gemm(16384, 3072, 768, arith='x6'), gemmf32=lambda: gemm(16384, 768, 2304),
           gemmbf16=lambda: gemm(16384, 768, 2304, arith='bf16'), gemmbf16_k3072=lambda: gemm(65536, 3072, 768, arith='bf16'),
           gemmbf16_big=lambda: gemm(65536, 768, 3072, 1, 'bf16'), gemmbf16_gelu=lambda: gemm(16384, 768, 3072, 1, 'bf16'),
           convin=convin, convin_x3h=lambda: convin(x3h=True), conv=conv, conv_nopro=lambda: conv(pro=False), gemm=gemm, gemm2=lambda: gemm(7168, 3072, 768),
           gemm_gelu=lambda: gemm(epi=1), gemm_tf=gemm_tf, x3h_stamps=x3h_stamps, convout=convout, attnsp=attnsp, attnsp_mid=lambda: attnsp(896, 64, 512), gemm_1x1=gemm_1x1, gemm_1x1_nin=lambda: gemm_1x1(917504, 128, 256, 4096, pro=False), gemm_1x1_mid=lambda: gemm_1x1(57344, 512, 1536, 64), g256_stamps=g256_stamps, g256_stamps_k768=lambda: g256_stamps(K=768, N=3072), g256_stamps_k128=lambda: g256_stamps(K=128, N=3072), gemm_tf_proj=lambda: gemm_tf(only='mlp.c_proj'), gemm_tf_fc=lambda: gemm_tf(only='mlp.c_fc'), gemm_tf_attn=lambda: gemm_tf(only='c_attn'), vq=vq, vq_bench=lambda: vq(64 * 896), vqf=vqf, vqf_stamps=vqf_stamps, vqf_small=lambda: vqf(64 * 56), vqf_big=lambda: vqf(64 * 8192), attn=attn, gn=gn,
           conv64=lambda: conv(56, 128, 64), conv256=lambda: conv(56, 256, 32), conv512=lambda: conv(224, 512, 8),
           convx6=lambda: conv(x6=True), convx6_64=lambda: conv(56, 128, 64, x6=True), convx6_256=lambda: conv(56, 256, 32, x6=True),
           convx6_512=lambda: conv(224, 512, 8, x6=True),
           convx3h=lambda: conv(x3h=True), convx3h_64=lambda: conv(56, 128, 64, x3h=True), convx3h_256=lambda: conv(56, 256, 32, x3h=True),
           convx3h_512=lambda: conv(224, 512, 8, x3h=True), convx3h_nopro=lambda: conv(pro=False, x3h=True))

if __name__ == '__main__':
    names = sys.argv[1:] or list(ALL)
    for n in names:
        ALL[n]()
