#!/usr/bin/env python
"""In-process A/B of the encoder's x3h convolutions across builds of libvf_hip.so: every library is loaded with its own handle and the SAME launch
is timed in alternation (box / minute clock drift cancels); outputs are compared bit for bit with the first library's.
  python tools/ab_inprocess_conv.py [--cases s1res,s1,s2,...] lib1.so lib2.so[:SEL=VAL] ...
A library given as path:SEL=VAL is loaded from a private COPY (its own vf_select state) with vf_select(SEL, VAL) applied: one build, two switch settings."""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from viewformer_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
args = sys.argv[1:]
cases = 's1res,s1,s2,s2_64,s2_32'
if args and args[0] == '--cases':
    cases, args = args[1], args[2:]
torch.zeros(1, device=dev)                                # (the HIP runtime is up before any library copy is loaded)


def _load(spec):
    if ':' not in spec:
        return os.path.basename(spec), _lib.load_variant(spec)
    import shutil
    import tempfile
    path, sel = spec.split(':')
    which, val = (int(x) for x in sel.split('='))
    tmp = os.path.join(tempfile.mkdtemp(prefix='vf_ab_'), f'libvf_sel{which}_{val}.so')
    shutil.copy(path, tmp)
    h = _lib.load_variant(tmp)
    assert h.vf_select(which, val) >= 0
    return f'{os.path.basename(path)}[vf_select({which},{val})]', h


libs = [_load(p) for p in args]


def bench(name, make):
    fn, out, flops = make()
    digests, times = {}, {n: [] for n, _ in libs}
    for n, h in libs:
        out.fill_(float('nan'))
        with _lib.use(h):
            fn()
        torch.cuda.synchronize()
        assert not torch.isnan(out).any(), (name, n)
        digests[n] = hash(out.view(torch.int32).cpu().numpy().tobytes())
    for r in range(10):
        for n, h in libs:
            with _lib.use(h):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / 6)
    med = {n: statistics.median(t) for n, t in times.items()}
    first = libs[0][0]
    print(json.dumps({'case': name, 'ms_median': {n: round(v, 4) for n, v in med.items()}, 'ms_min': {n: round(min(t), 4) for n, t in times.items()},
                      'tflops_fp32_equiv': {n: round(flops / v / 1e9, 1) for n, v in med.items()},
                      'vs_first': {n: round(v / med[first], 4) for n, v in med.items()},
                      'same_bits_as_first': {n: digests[n] == digests[first] for n in digests}}), flush=True)


def s1(n_img=448, C=128, H=128, res=True, pro=True):
    def make():
        g = torch.Generator().manual_seed(1)
        x = torch.randn(n_img * H * H, C, generator=g).to(dev)
        w = (torch.randn(C, C, 3, 3, generator=g) * 0.03).to(dev)
        wp = ops.pack_conv3_x3h(w)
        b = torch.randn(C, generator=g).to(dev)
        r = torch.randn(n_img * H * H, C, generator=g).to(dev) if res else None
        out = torch.empty_like(x)
        prol = None
        if pro:
            m, s_ = ops.groupnorm_stats(x, torch.ones(C, device=dev), n_img, H * H, C)
            prol = (m, s_, torch.zeros(C, device=dev))
        part = ops.new_gn_part(n_img, H, H, dev)
        M = n_img * H * H
        return (lambda: ops.igemm(x, wp, M, C, C, out, bias=b, res=r, mode=ops.MODE_CONV3_S1, pro=prol, pro_swish=True, pro_rows_per_img=H * H,
                                  Hin=H, Win=H, Hout=H, Wout=H, x3h=True, gn_part=part)), out, 2.0 * M * C * C * 9
    return make


def s2(n_img=448, C=128, H=128):
    def make():
        g = torch.Generator().manual_seed(2)
        x = torch.randn(n_img * H * H, C, generator=g).to(dev)
        w = (torch.randn(C, C, 3, 3, generator=g) * 0.03).to(dev)
        wp = ops.pack_conv3_x3h(w)
        b = torch.randn(C, generator=g).to(dev)
        Ho = H // 2
        M = n_img * Ho * Ho
        out = torch.empty(M, C, device=dev)
        part = ops.new_gn_part(n_img, Ho, Ho, dev)
        return (lambda: ops.igemm(x, wp, M, C, C, out, bias=b, mode=ops.MODE_CONV3_S2PAD, Hin=H, Win=H, Hout=Ho, Wout=Ho, x3h=True, gn_part=part)), out, \
            2.0 * M * C * C * 9
    return make


ALL = {'s1res': ('3x3 s1 128->128 @128^2 x448, GN+swish prologue, residual (ResnetBlock conv2)', s1()),
       's1': ('3x3 s1 128->128 @128^2 x448, GN+swish prologue, no residual (conv1)', s1(res=False)),
       's1res64': ('3x3 s1 128->128 @64^2 x896, GN+swish, residual', s1(896, 128, 64)),
       's1res256': ('3x3 s1 256->256 @32^2 x896, GN+swish, residual', s1(896, 256, 32)),
       's2': ('3x3 s2 128->128 @128^2 -> 64^2 x448 (Downsample 0)', s2()),
       's2_64': ('3x3 s2 128->128 @64^2 -> 32^2 x896 (Downsample 1)', s2(896, 128, 64)),
       's2_32': ('3x3 s2 256->256 @32^2 -> 16^2 x896 (Downsample 2)', s2(896, 256, 32))}
for c in cases.split(','):
    bench(*ALL[c])
