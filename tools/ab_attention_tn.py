#!/usr/bin/env python
"""A/B of the kernels that read LDS with ds_read_b64_tr_b16 (attention forward, attention backward, TN weight-gradient GEMM): output
digests (bit-identity across two builds of the library: run once per VF_HIP_LIB and diff the 'digest' fields) and times.
  VF_HIP_LIB=<lib> python tools/ab_attention_tn.py [tag]      -> one JSON line"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from viewformer_amd import ops  # noqa: E402
from viewformer_amd import train_ops as T  # noqa: E402

dev = torch.device('cuda:0')


def digest(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    res = {'tag': sys.argv[1] if len(sys.argv) > 1 else '', 'lib': os.environ.get('VF_HIP_LIB', 'default')}
    g = torch.Generator(device='cpu').manual_seed(7)
    # ---- inference attention, bench shape: 128 scenes x 12 heads x 512 tokens, fused twin mask
    B, H, S, L = 128, 12, 8, 64
    d, Tn = H * 64, S * L
    qkv = (torch.randn(B * Tn, 3 * d, generator=g) * 0.3).to(dev).to(torch.bfloat16)
    out = torch.empty(B * Tn, d, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.attn_blockcausal(qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d], out, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, True, 6, bf16=True)  # noqa: E731
    f()
    res['attn_fwd_bench'] = {'digest': digest(out), 'us': round(timeit(f), 1)}
    # ---- S = 20 localization shape (configs[2])
    B2, S2 = 45, 21
    T2 = S2 * L
    qkv2 = (torch.randn(B2 * T2, 3 * d, generator=g) * 0.3).to(dev).to(torch.bfloat16)
    out2 = torch.empty(B2 * T2, d, device=dev, dtype=torch.bfloat16)
    f2 = lambda: ops.attn_blockcausal(qkv2[:, d:2 * d], qkv2[:, 2 * d:], qkv2[:, :d], out2, B2, H, T2, L, 3 * d, 3 * d, 3 * d, d, 1.0, True, 19, bf16=True)  # noqa: E731
    f2()
    res['attn_fwd_s20'] = {'digest': digest(out2), 'us': round(timeit(f2), 1)}
    # ---- training shape: 10 scenes x 12 heads x 1920 tokens, 3-stream mask, dropout 0.1 and 0
    B3, S3 = 10, 30
    T3 = S3 * L
    qkv3 = (torch.randn(B3 * T3, 3 * d, generator=g) * 0.3).to(dev).to(torch.bfloat16)
    o3 = torch.empty(B3 * T3, d, device=dev, dtype=torch.bfloat16)
    do3 = (torch.randn(B3 * T3, d, generator=g) * 0.1).to(dev).to(torch.bfloat16)
    for rate in (0.0, 0.1):
        drop = (rate, 17, 5)
        q, k, v = qkv3[:, d:2 * d], qkv3[:, 2 * d:], qkv3[:, :d]
        lse = T.attn_fwd_lse_bf16(q, k, v, o3, B3, H, T3, L, 3 * d, 3 * d, 3 * d, d, 1.0, -10, drop)
        dqkv = torch.zeros(B3 * T3, 3 * d, device=dev, dtype=torch.bfloat16)
        fb = lambda: T.attn_bwd_bf16(q, k, v, o3, do3, lse, dqkv[:, d:2 * d], dqkv[:, 2 * d:], dqkv[:, :d], B3, H, T3, L, 3 * d, 3 * d, 3 * d, d, d,  # noqa: E731
                                     3 * d, 3 * d, 3 * d, 1.0, -10, drop)
        fb()
        ff = lambda: T.attn_fwd_lse_bf16(q, k, v, o3, B3, H, T3, L, 3 * d, 3 * d, 3 * d, d, 1.0, -10, drop)  # noqa: E731
        res[f'attn_train_drop{rate}'] = {'digest_fwd': digest(o3, lse), 'digest_bwd': digest(dqkv), 'fwd_us': round(timeit(ff), 1),
                                         'bwd_us': round(timeit(fb), 1)}
    # ---- TN weight-gradient GEMM: the four layer shapes at M = 19 200, fp32 and bf16 dY
    for K, N in ((768, 2304), (768, 768), (768, 3072), (3072, 768)):
        M = 19200
        x16 = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(torch.bfloat16)
        for y16 in (False, True):
            dy = (torch.randn(M, N, generator=g) * 0.1).to(dev)
            if y16:
                dy = dy.to(torch.bfloat16)
            if not ops.gemm_tn_bf16_supported(x16, M, K, N):
                continue
            dw = torch.zeros(K, N, device=dev)
            db = torch.zeros(N, device=dev)
            ops.gemm_tn_bf16(x16, dy, M, K, N, dw, db, accumulate=False)
            us = timeit(lambda: ops.gemm_tn_bf16(x16, dy, M, K, N, dw, db, accumulate=False))
            res[f'tn_{K}x{N}_{"y16" if y16 else "y32"}'] = {'digest': digest(dw, db), 'us': round(us, 1), 'tflops': round(2.0 * M * K * N / us / 1e6, 1)}
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
