"""GPU: the fp32-equivalent split-bf16 ("x6") convolution arm.  The claim under test: its error against an fp64
reference is no larger than the native f32-MFMA kernel's on the same inputs (it is NOT a reduced-precision arm),
for every geometry (plain 8x16 tiles, nearest-x2 upsample, 8x8 pair tiles, odd image counts) with and without the
fused GroupNorm+swish prologue and the residual."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs the MI355X'
    return torch.device('cuda:0')


def _rand(shape, seed, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def _err(out, ref, mag):
    """max and rms of |out - ref| / (sum of |products| of that output): the rounding-error scale of a dot product"""
    e = (out.double().cpu() - ref).abs() / mag
    return e.max().item(), e.pow(2).mean().sqrt().item()


@pytest.mark.parametrize('mode,cin,cout,H,pro', [('s1', 128, 128, 16, True), ('s1', 64, 256, 32, False), ('up', 128, 128, 8, False),
                                                  ('up', 32, 128, 16, True), ('s1', 128, 128, 64, True),
                                                  ('s1', 128, 128, 8, True), ('s1', 512, 512, 8, False), ('s1', 256, 256, 16, True)])
def test_conv3_halo_x6_is_fp32_equivalent(dev, mode, cin, cout, H, pro):
    from viewformer_amd import ops
    n = 3 if H == 8 else 2
    x = _rand((n, cin, H, H), 11) * 1.5 + 0.2
    w, b = _rand((cout, cin, 3, 3), 12, 0.05), _rand((cout,), 13)
    gamma, beta = _rand((cin,), 14) * 0.3 + 1, _rand((cin,), 15) * 0.2
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
    prol, a = None, x.double()
    if pro:
        mean_c, scale_c = ops.groupnorm_stats(xn, gamma.to(dev), n, H * H, cin)
        prol = (mean_c, scale_c, beta.to(dev))
        # the reference applies the SAME fp32 statistics in fp64 so that only the convolution arithmetic is compared
        mu = mean_c.double().cpu().view(n, cin, 1, 1)
        sc = scale_c.double().cpu().view(n, cin, 1, 1)
        a = (a - mu) * sc + beta.double().view(1, cin, 1, 1)
        a = a * torch.sigmoid(a)
    m, Ho = (ops.MODE_CONV3_S1, H) if mode == 's1' else (ops.MODE_CONV3_UP2, 2 * H)
    if mode == 'up':
        a = F.interpolate(a, scale_factor=2.0, mode='nearest')
    ref = F.conv2d(a, w.double(), None, padding=1)
    mag = F.conv2d(a.abs(), w.double().abs(), None, padding=1)
    res = _rand((n * Ho * Ho, cout), 16)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, cout) + b.double() + res.double()
    mag = mag.permute(0, 2, 3, 1).reshape(-1, cout) + b.double().abs() + res.double().abs()
    kw = dict(bias=b.to(dev), res=res.to(dev), mode=m, pro=prol, pro_swish=True, Hin=H, Win=H, Hout=Ho, Wout=Ho)
    assert ops.conv3_x6_supported(m, cin, cout, Ho, Ho)
    o6 = torch.empty((n * Ho * Ho, cout), device=dev)
    ops.igemm(xn, ops.pack_conv3_x6(w.to(dev)), n * Ho * Ho, cin, cout, o6, x6=True, **kw)
    o32 = torch.empty((n * Ho * Ho, cout), device=dev)
    ops.igemm(xn, ops.pack_conv_oihw(w.to(dev)), n * Ho * Ho, cin, cout, o32, **kw)
    (mx6, rms6), (mx32, rms32) = _err(o6, ref, mag), _err(o32, ref, mag)
    print(f'{mode} {cin}->{cout} @{H} pro={pro}: x6 max {mx6:.2e} rms {rms6:.2e} | f32 MFMA max {mx32:.2e} rms {rms32:.2e}')
    # fp32-level absolute bar, and "no worse than the native f32 kernel" (25 % slack on the rms for sampling noise)
    assert mx6 < 6e-7 and rms6 < 1.25 * rms32 + 1e-9


def test_x6_refuses_unsupported_shapes(dev):
    from viewformer_amd import ops
    x = torch.zeros((2 * 16 * 12, 64), device=dev)
    w = ops.pack_conv3_x6(torch.zeros((128, 64, 3, 3), device=dev))
    out = torch.empty((2 * 16 * 12, 128), device=dev)
    with pytest.raises(ops._lib.VfError):        # W % 16 != 0: refused, never rerouted
        ops.igemm(x, w, 2 * 16 * 12, 64, 128, out, mode=ops.MODE_CONV3_S1, Hin=16, Win=12, Hout=16, Wout=12, x6=True)
    assert not ops.conv3_x6_supported(ops.MODE_CONV3_S2PAD, 128, 128, 64, 64)
