"""GPU: the DMA-ring kernels whose transposing LDS reads are inline asm (csrc/vf_common.h: vf_tr_frag2_wait) against a build that issues the
same reads through the compiler intrinsic (-DVF_X_TRINTRIN: hipcc then orders every read behind the ring with its own ``s_waitcnt vmcnt(0)``),
RESULT for result (until round 6 the equivalence was only checked at the ISA level, tests/test_isa_audit.py).

The inline-asm form hides the reads from the compiler's waitcnt pass, so the ring's correctness hangs on the kernels' hand-counted vmcnt waits
plus the barrier at the top of every ring step; a miscount would be a rare, load-dependent race — wrong tiles that come and go.  The stress
runs each kernel many times on fresh inputs while a second stream keeps the memory system busy (DMA landing times move), and every output of
the product build must equal the compiler-ordered build's bit for bit."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def libs():
    from viewformer_amd import _lib, build
    path = build.variant_path('trintrin')
    if not os.path.exists(path):                     # (hipcc is on the GPU box too; __graft_entry__.build() builds it beforehand)
        path = build.build_variant('trintrin')
    ref = _lib.load_variant(path)
    names = [ref.vf_build_flag_name(i).decode() for i in range(ref.vf_build_flags())]
    assert 'VF_X_TRINTRIN' in names, names           # the reference really is the intrinsic build
    assert _lib.load().vf_build_flags() == 0         # and the product build carries no developer flag
    return _lib, ref


class _Noise:
    """a second stream that streams a few hundred MB through HBM / L2 while the kernel under test runs"""

    def __init__(self, dev):
        self.s = torch.cuda.Stream(dev)
        self.a = torch.empty(64 << 20, dtype=torch.float32, device=dev)
        self.b = torch.empty_like(self.a)

    def kick(self, n):
        with torch.cuda.stream(self.s):
            for _ in range(n):
                self.b.copy_(self.a)


def _attn_inputs(B, H, S, seed, dev, scale=0.5):
    d, T = H * 64, S * 64
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(B * T, 3 * d, generator=g) * scale).to(dev).to(torch.bfloat16)
    return qkv, d, T


@pytest.mark.parametrize('B,H,S,twin', [(8, 12, 8, 6), (3, 12, 21, 19), (5, 4, 3, -1)])
def test_attention_forward_ring_equals_the_compiler_ordered_build(libs, B, H, S, twin):
    _lib, ref = libs
    from viewformer_amd import ops
    dev = torch.device('cuda:0')
    noise = _Noise(dev)
    for it in range(12):
        qkv, d, T = _attn_inputs(B, H, S, 100 + it, dev)
        q, k, v = qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d]
        outs = []
        for lib in (None, ref, None):
            o = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
            noise.kick(it % 3)
            if lib is None:
                ops.attn_blockcausal(q, k, v, o, B, H, T, 64, 3 * d, 3 * d, 3 * d, d, 1.0, True, twin, bf16=True)
            else:
                with _lib.use(lib):
                    ops.attn_blockcausal(q, k, v, o, B, H, T, 64, 3 * d, 3 * d, 3 * d, d, 1.0, True, twin, bf16=True)
            outs.append(o)
        torch.cuda.synchronize()
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), f'iteration {it}: inline-asm build != intrinsic build'
        assert torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16)), f'iteration {it}: the product build is not reproducible'


@pytest.mark.parametrize('drop', [0.0, 0.1])
def test_training_attention_rings_equal_the_compiler_ordered_build(libs, drop):
    _lib, ref = libs
    from viewformer_amd import train_ops as T_
    dev = torch.device('cuda:0')
    noise = _Noise(dev)
    B, H, S = 4, 12, 30                                       # the training step's shape class: 3 streams x 10 views
    for it in range(8):
        qkv, d, T = _attn_inputs(B, H, S, 300 + it, dev)
        q, k, v = qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d]
        g = torch.Generator().manual_seed(900 + it)
        dout = (torch.randn(B * T, d, generator=g) * 0.1).to(dev).to(torch.bfloat16)
        res = []
        for lib in (None, ref):
            noise.kick(it % 3)
            o = torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)
            dqkv = torch.zeros(B * T, 3 * d, device=dev, dtype=torch.bfloat16)

            def run():
                lse = T_.attn_fwd_lse_bf16(q, k, v, o, B, H, T, 64, 3 * d, 3 * d, 3 * d, d, 1.0, -10, (drop, 7 + it, 2))
                T_.attn_bwd_bf16(q, k, v, o, dout, lse, dqkv[:, d:2 * d], dqkv[:, 2 * d:], dqkv[:, :d], B, H, T, 64, 3 * d, 3 * d, 3 * d, d, d,
                                 3 * d, 3 * d, 3 * d, 1.0, -10, (drop, 7 + it, 2))
                return lse
            if lib is None:
                lse = run()
            else:
                with _lib.use(lib):
                    lse = run()
            res.append((o, lse, dqkv))
        torch.cuda.synchronize()
        for name, a, b in zip(('out', 'lse', 'dqkv'), res[0], res[1]):
            assert torch.equal(a.view(torch.int16 if a.dtype == torch.bfloat16 else torch.int32),
                               b.view(torch.int16 if b.dtype == torch.bfloat16 else torch.int32)), f'iteration {it}: {name} differs'


@pytest.mark.parametrize('M,K,N,y16', [(19200, 768, 3072, False), (19200, 3072, 768, True), (4096, 768, 768, False)])
def test_weight_gradient_ring_equals_the_compiler_ordered_build(libs, M, K, N, y16):
    _lib, ref = libs
    from viewformer_amd import ops
    dev = torch.device('cuda:0')
    noise = _Noise(dev)
    for it in range(6):
        g = torch.Generator().manual_seed(500 + it)
        x = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
        dy = (torch.randn(M, N, generator=g) * 0.1).to(dev)
        if y16:
            dy = dy.to(torch.bfloat16)
        res = []
        for lib in (None, ref):
            noise.kick(it % 3)
            dw = torch.zeros(K * N + N, device=dev)
            if lib is None:
                ops.gemm_tn_bf16(x, dy, M, K, N, dw[:K * N].view(K, N), dw[K * N:], accumulate=False)
            else:
                with _lib.use(lib):
                    ops.gemm_tn_bf16(x, dy, M, K, N, dw[:K * N].view(K, N), dw[K * N:], accumulate=False)
            res.append(dw)
        torch.cuda.synchronize()
        assert torch.equal(res[0].view(torch.int32), res[1].view(torch.int32)), f'iteration {it}: dW / db differ'
