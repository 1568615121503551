#!/usr/bin/env python
"""In-process A/B of the 256-tile bf16 GEMM's tail policy (VF_SEL_GEMM_TAIL, round 6) at the transformer's shapes: the same launch with the
switch off / on in alternation (clock drift cancels), bit-identity of the two outputs, and optionally other builds of the library beside them
(e.g. round 5's kernel: python tools/ab_gemm_tail.py viewformer_amd/variants/libvf_g256_r5.so)."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from viewformer_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
others = [(os.path.basename(p), _lib.load_variant(p)) for p in sys.argv[1:]]


def case(name, M, K, N, kind):
    x16 = torch.randn((M, K), generator=g).to(dev).to(torch.bfloat16)
    wp = ops.pack_dense_kn_bf16((torch.randn((K, N), generator=g) * 0.05).to(dev))
    b = torch.randn(N, generator=g).to(dev)
    f32out = kind in ('res32', 'res32_drop')
    o16 = None if f32out else torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    f16 = torch.empty((M, N), dtype=torch.bfloat16, device=dev) if kind == 'dual' else None
    o32 = torch.empty((M, N), device=dev) if f32out else None
    res = torch.randn((M, N), generator=g).to(dev) if f32out else None
    u16 = torch.randn((M, N), generator=g).to(dev).to(torch.bfloat16) if kind == 'gelu_bwd' else None

    def fn():
        if kind == 'bf16':
            ops.igemm(x16, wp, M, K, N, o16, bias=b, bf16=True, a16=True, o16=True)
        elif kind == 'dual':
            ops.igemm(x16, wp, M, K, N, o16, bias=b, epilogue=ops.EPI_GELU_DUAL, bf16=True, a16=True, o16=True, out_aux=f16)
        elif kind == 'gelu':
            ops.igemm(x16, wp, M, K, N, o16, bias=b, epilogue=ops.EPI_GELU, bf16=True, a16=True, o16=True)
        elif kind == 'gelu_bwd':
            ops.igemm(x16, wp, M, K, N, o16, res=u16, epilogue=ops.EPI_GELU_BWD, bf16=True, a16=True, o16=True, res16=True)
        elif kind == 'res32_drop':
            ops.igemm(x16, wp, M, K, N, o32, bias=b, res=res, bf16=True, a16=True, drop=(0.1, 5, 1))
        elif kind == 'res32':
            ops.igemm(x16, wp, M, K, N, o32, bias=b, res=res, bf16=True, a16=True)
        return o16 if o16 is not None else o32
    arms = [('tail_off', None, 0), ('tail_on', None, 1)] + [(n, h, 0) for n, h in others]
    digests, times = {}, {n: [] for n, _, _ in arms}

    def run(arm, reps):
        n, h, sel = arm
        prev = _lib.select(_lib.SEL_GEMM_TAIL, sel)
        try:
            if h is None:
                for _ in range(reps):
                    out = fn()
            else:
                with _lib.use(h):
                    for _ in range(reps):
                        out = fn()
        finally:
            _lib.select(_lib.SEL_GEMM_TAIL, prev)
        return out
    for arm in arms:
        out = run(arm, 1)
        torch.cuda.synchronize()
        digests[arm[0]] = hash(out.view(torch.int16 if out.dtype == torch.bfloat16 else torch.int32).cpu().numpy().tobytes())
    for r in range(10):
        for arm in arms:
            run(arm, 3)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(arm, 20)
            e1.record()
            torch.cuda.synchronize()
            times[arm[0]].append(e0.elapsed_time(e1) / 20 * 1e3)
    tiles = -(-M // 256) * (N // 256)
    med = {n: statistics.median(t) for n, t in times.items()}
    print(json.dumps({'case': name, 'M_K_N': [M, K, N], 'epilogue': kind, 'tiles': tiles, 'rounds_of_256': round(tiles / 256, 3),
                      'us_median': {n: round(v, 1) for n, v in med.items()}, 'us_min': {n: round(min(t), 1) for n, t in times.items()},
                      'tflops_median': {n: round(2.0 * M * K * N / v / 1e6) for n, v in med.items()},
                      'tail_on_vs_off': round(med['tail_on'] / med['tail_off'], 4),
                      'same_bits_as_tail_off': {n: digests[n] == digests['tail_off'] for n in digests}}), flush=True)


# training step (M = 10 scenes x 3 streams x 640 tokens)
case('train c_attn', 19200, 768, 2304, 'bf16')
case('train c_fc (GELU dual)', 19200, 768, 3072, 'dual')
case('train mlp.c_proj^T dX (GELU backward)', 19200, 768, 3072, 'gelu_bwd')
case('train attn.c_proj (fp32 out + residual + dropout)', 19200, 768, 768, 'res32_drop')
case('train mlp.c_proj (fp32 out + residual + dropout)', 19200, 3072, 768, 'res32_drop')
case('train c_attn^T dX', 19200, 2304, 768, 'bf16')
# inference (M = 128 scenes x 7 views x 64 tokens)
case('infer c_attn', 57344, 768, 2304, 'bf16')
case('infer c_fc (GELU)', 57344, 768, 3072, 'gelu')
case('infer mlp.c_proj', 57344, 3072, 768, 'res32')
case('infer attn.c_proj', 57344, 768, 768, 'res32')
# reference points: whole rounds
case('M = 65536 c_fc (12 rounds)', 65536, 768, 3072, 'bf16')
