#!/usr/bin/env python
"""In-process A/B of the 256-tile bf16 GEMM across builds of the library at the transformer's shapes (training M = 19 200 / 12 800, inference M = 65 536), launches
alternated, outputs compared bit for bit with the first library's.  usage: python tools/ab_gemm_libs.py lib1.so lib2.so ..."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from viewformer_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
libs = [(os.path.basename(p), _lib.load_variant(p)) for p in sys.argv[1:]]
g = torch.Generator(device='cpu').manual_seed(0)
for (M, K, N, o16) in ((19200, 768, 3072, True), (19200, 3072, 768, False), (19200, 768, 2304, True), (19200, 768, 768, False), (12800, 768, 3072, True),
                       (65536, 768, 3072, True), (65536, 3072, 768, False), (65536, 768, 2304, True)):
    x16 = torch.randn((M, K), generator=g).to(dev).to(torch.bfloat16)
    with _lib.use(libs[0][1]):
        wp = ops.pack_dense_kn_bf16((torch.randn((K, N), generator=g) * 0.05).to(dev))
    outs, times = {}, {n: [] for n, _ in libs}
    for r in range(8):
        for n, h in (libs if r % 2 == 0 else libs[::-1]):
            with _lib.use(h):
                o = torch.empty((M, N), dtype=torch.bfloat16 if o16 else torch.float32, device=dev)
                fn = lambda: ops.igemm(x16, wp, M, K, N, o, bf16=True, a16=True, o16=o16)      # noqa: E731
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) * 1000 / 20)
                outs[n] = o
    print(json.dumps({'M_K_N_out16': [M, K, N, o16], 'us_median': {n: round(statistics.median(t), 2) for n, t in times.items()},
                      'same_bits_as_first': {n: bool(torch.equal(outs[n], outs[libs[0][0]])) for n in outs}}), flush=True)
