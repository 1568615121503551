"""GPU: run-to-run bit reproducibility WHILE ANOTHER PROCESS SHARES THE GPU (round 3 found the two-rows-in-flight LayerNorm backward returning a
few dx rows ~1e-4 off about once in 75 calls under exactly this condition; ADVICE r3 asks for the probe to stay in the suite, for that kernel and for
kernels that still reduce with vf_wave_sum / __shfl_xor).  Every call must return the bits of the first call."""
import os
import subprocess
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEIGHBOUR = r'''
import sys, time, torch
dev = torch.device('cuda:0')
a = torch.randn((4096, 4096), device=dev)
x = torch.randn((1 << 24,), device=dev)
print('up', flush=True)
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(20):
        b = a @ a
        x = x * 1.0001 + 0.5
    torch.cuda.synchronize()
'''


@pytest.fixture(scope='module')
def neighbour():
    """a second process that keeps cuda:0 busy with GEMMs and streaming kernels for the duration of the module"""
    p = subprocess.Popen([sys.executable, '-c', NEIGHBOUR, '60'], stdout=subprocess.PIPE, text=True, cwd=ROOT)
    assert p.stdout.readline().strip() == 'up'
    yield p
    p.kill()
    p.wait()


def _repeat(call, n, what):
    ref = [t.clone() for t in call()]
    torch.cuda.synchronize()
    filler = torch.randn((2048, 2048), device='cuda:0')
    bad = 0
    for _ in range(n):
        _ = filler @ filler                                        # other work between the calls, as in a step
        out = call()
        torch.cuda.synchronize()
        bad += any(not torch.equal(a, b) for a, b in zip(out, ref))
    assert bad == 0, f'{what}: {bad} of {n} calls differ from the first'


def test_layernorm_backward_is_bit_reproducible_on_a_shared_gpu(neighbour):
    from viewformer_amd import train_ops as T
    dev = torch.device('cuda:0')
    M, d = 3840, 768
    g = torch.Generator().manual_seed(0)
    dy, x, res = (torch.randn((M, d), generator=g).to(dev) for _ in range(3))
    gamma = torch.randn(d, generator=g).to(dev)

    def call():
        dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
        dx, dx16 = T.layernorm_bwd(dy, x, gamma, dg, db, M, d, res=res, also_bf16=True, drop=(0.1, 7, 3, 0))
        return dx, dx16, dg, db
    assert neighbour.poll() is None
    _repeat(call, 150, 'layernorm_bwd (+ masked bf16 copy)')


def test_wave_sum_reductions_are_bit_reproducible_on_a_shared_gpu(neighbour):
    """kernels that reduce with vf_wave_sum (__shfl_xor): LayerNorm forward, softmax cross-entropy, GroupNorm statistics"""
    from viewformer_amd import ops
    from viewformer_amd import train_ops as T
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1)
    M, d, V = 3840, 768, 1024
    x = torch.randn((M, d), generator=g).to(dev)
    gamma, beta = torch.randn(d, generator=g).to(dev), torch.randn(d, generator=g).to(dev)
    logits = (torch.randn((M, V), generator=g) * 3).to(dev)
    target = torch.randint(0, V, (M,), generator=g).to(torch.int32).to(dev)
    weight = torch.rand(M, generator=g).to(dev)
    n_img, HW, C = 8, 4096, 128
    xg = torch.randn((n_img * HW, C), generator=g).to(dev)
    gg = torch.randn(C, generator=g).to(dev)

    def call():
        y = ops.layernorm(x, gamma, beta, M, d)
        y16 = ops.layernorm(x, gamma, beta, M, d, out_bf16=True)
        loss, dl = T.softmax_ce(logits, target, weight, M, V, 0.1)[:2]
        mean_c, scale_c = ops.groupnorm_stats(xg, gg, n_img, HW, C)
        return y, y16, loss, dl, mean_c, scale_c
    assert neighbour.poll() is None
    _repeat(call, 100, 'layernorm / softmax_ce / groupnorm_stats')
