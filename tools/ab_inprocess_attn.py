#!/usr/bin/env python
"""In-process A/B of the LDS-DMA attention across several builds of libvf_hip.so: every library is loaded with its own ctypes handle and
the SAME launch (bench shape: 128 scenes x 12 heads x 512 tokens, fused twin mask, bf16 in / out) is timed in alternation — box-to-box and
minute-to-minute clock drift (+-5 % between two processes on one box) cancels.  usage: python tools/ab_inprocess_attn.py lib1.so lib2.so ..."""
import ctypes
import json
import statistics
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

P, c_int, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
dev = torch.device('cuda:0')
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    f = lib.vf_attn_blockcausal_bf16_v2
    f.restype, f.argtypes = c_int, [P, P, P, c_int, P, c_int] + [c_int] * 8 + [c_float, c_int, c_int, P]
    libs.append((os.path.basename(path), f))


def run(shape_name, B, H, S, twin, rounds=12, iters=25):
    L = 64
    d, T = H * 64, S * L
    g = torch.Generator().manual_seed(3)
    qkv = (torch.randn(B * T, 3 * d, generator=g) * 0.3).to(dev).to(torch.bfloat16)
    out = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
    q, k, v = qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(f):
        rc = f(P(q.data_ptr()), P(k.data_ptr()), P(v.data_ptr()), 1, P(out.data_ptr()), 1, B, H, T, L, 3 * d, 3 * d, 3 * d, d, 1.0, 1, twin, st)
        assert rc == 0, rc
    times = {n: [] for n, _ in libs}
    digests = {}
    for n, f in libs:
        call(f)
        torch.cuda.synchronize()
        digests[n] = hash(out.view(torch.int16).cpu().numpy().tobytes())
    for r in range(rounds):
        for n, f in libs:
            for _ in range(3):
                call(f)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                call(f)
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / iters * 1e3)
    ref = digests[libs[0][0]]
    print(json.dumps({'shape': shape_name, 'B_H_S_twin': [B, H, S, twin],
                      'us_median': {n: round(statistics.median(t), 1) for n, t in times.items()},
                      'us_min': {n: round(min(t), 1) for n, t in times.items()},
                      'same_bits_as_first': {n: digests[n] == ref for n in digests}}), flush=True)


run('bench (6 context views + MASK + LOC view)', 128, 12, 8, 6)
run('configs[2] S = 21 views', 45, 12, 21, 19)
