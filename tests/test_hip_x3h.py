"""GPU: the fp32-equivalent split-fp16 ("x3h") convolution — two fp16 pieces per operand with the low piece carried at 2^11,
three products, cross terms in their own accumulator, power-of-two pre-scaled weights (csrc/conv3_halo_x3h.hip).  The claim under
test: for activations inside fp16's range its error against fp64 is no larger than the native f32-MFMA kernel's (it is NOT a
reduced-precision arm); outside that range it degrades as documented, which is why x6 stays the arithmetic of the backward pass."""
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
    e = (out.double().cpu() - ref).abs() / mag
    return e.max().item(), e.pow(2).mean().sqrt().item()


def _run(dev, mode, cin, cout, H, pro, xscale=1.5, wscale=0.05, n=None, swish=True):
    from viewformer_amd import ops
    n = n or (3 if H == 8 else 2)
    x = _rand((n, cin, H, H), 11) * xscale + 0.2 * xscale
    w, b = _rand((cout, cin, 3, 3), 12, wscale), _rand((cout,), 13) * wscale * xscale * 10
    gamma, beta = _rand((cin,), 14) * 0.3 + 1, _rand((cin,), 15) * 0.2
    xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
    prol, a = None, x.double()
    if pro:
        mean_c, scale_c = ops.groupnorm_stats(xn, gamma.to(dev), n, H * H, cin)
        prol = (mean_c, scale_c, beta.to(dev))
        mu = mean_c.double().cpu().view(n, cin, 1, 1)
        sc = scale_c.double().cpu().view(n, cin, 1, 1)
        a = (a - mu) * sc + beta.double().view(1, cin, 1, 1)
        if swish:
            a = a * torch.sigmoid(a)
    m, Ho = {'s1': (ops.MODE_CONV3_S1, H), 'up': (ops.MODE_CONV3_UP2, 2 * H), 's2': (ops.MODE_CONV3_S2PAD, H // 2)}[mode]
    if mode == 'up':
        a = F.interpolate(a, scale_factor=2.0, mode='nearest')
    if mode == 's2':                                      # Downsample: pad (right, bottom) + stride 2
        ap = F.pad(a, (0, 1, 0, 1))
        ref = F.conv2d(ap, w.double(), None, stride=2)
        mag = F.conv2d(ap.abs(), w.double().abs(), None, stride=2)
    else:
        ref = F.conv2d(a, w.double(), None, padding=1)
        mag = F.conv2d(a.abs(), w.double().abs(), None, padding=1)
    res = _rand((n * Ho * Ho, cout), 16) * wscale * xscale * 10
    ref = ref.permute(0, 2, 3, 1).reshape(-1, cout) + b.double() + res.double()
    mag = mag.permute(0, 2, 3, 1).reshape(-1, cout) + b.double().abs() + res.double().abs()
    kw = dict(bias=b.to(dev), res=res.to(dev), mode=m, pro=prol, pro_swish=swish, Hin=H, Win=H, Hout=Ho, Wout=Ho)
    assert ops.conv3_x3h_supported(m, cin, cout, Ho, Ho)
    o3 = torch.empty((n * Ho * Ho, cout), device=dev)
    ops.igemm(xn, ops.pack_conv3_x3h(w.to(dev)), n * Ho * Ho, cin, cout, o3, x3h=True, **kw)
    o32 = torch.empty((n * Ho * Ho, cout), device=dev)
    ops.igemm(xn, ops.pack_conv_oihw(w.to(dev)), n * Ho * Ho, cin, cout, o32, **kw)
    return _err(o3, ref, mag), _err(o32, ref, mag)


def test_conv3_s2_x3h_pair_form_odd_image_count(dev):
    """16x16 -> 8x8 Downsample: two images per tile; an odd image count processes the last image twice (same values rewritten)"""
    (mx3, rms3), (mx32, rms32) = _run(dev, 's2', 128, 128, 16, False, n=3)
    assert mx3 < 6e-7 and rms3 < 1.25 * rms32 + 1e-9, (mx3, rms3, rms32)


@pytest.mark.parametrize('mode,cin,cout,H,pro', [('s1', 128, 128, 16, True), ('s1', 64, 256, 32, False), ('up', 128, 128, 8, False),
                                                  ('up', 32, 128, 16, True), ('s1', 128, 128, 64, True),
                                                  ('s1', 128, 128, 8, True), ('s1', 512, 512, 8, False), ('s1', 256, 256, 16, True),
                                                  ('s2', 128, 128, 32, False), ('s2', 64, 256, 64, False), ('s2', 32, 128, 32, False),
                                                      ('s2', 256, 256, 16, False), ('s2', 64, 128, 16, False)])
def test_conv3_halo_x3h_is_fp32_equivalent(dev, mode, cin, cout, H, pro):
    (mx3, rms3), (mx32, rms32) = _run(dev, mode, cin, cout, H, pro)
    print(f'{mode} {cin}->{cout} @{H} pro={pro}: x3h max {mx3:.2e} rms {rms3:.2e} | f32 MFMA max {mx32:.2e} rms {rms32:.2e}')
    assert mx3 < 6e-7 and rms3 < 1.25 * rms32 + 1e-9


@pytest.mark.parametrize('cin,cout,H,pro', [(128, 128, 16, True), (64, 256, 32, False), (128, 128, 64, True), (256, 256, 16, True), (512, 512, 16, False),
                                            (512, 128, 16, True)])
def test_conv3_halo_x3h_both_mfma_shapes_meet_the_same_bound(dev, cin, cout, H, pro):
    """stride-1 halo convolution on v_mfma_f32_16x16x32_f16 (default where it applies: even chunk count, LDS for two workgroups) and on
    32x32x16 (vf_select(VF_SEL_CONV_X3H_K32, 0)): another accumulation order -> different last bits, the SAME fp32-equivalence bound; the fused
    GroupNorm partials of both describe what was stored"""
    from viewformer_amd import ops, _lib
    res = {}
    try:
        for k32 in (1, 0):
            _lib.select(_lib.SEL_CONV_X3H_K32, k32)
            res[k32] = _run(dev, 's1', cin, cout, H, pro)
    finally:
        _lib.select(_lib.SEL_CONV_X3H_K32, 1)
    for k32, ((mx3, rms3), (mx32, rms32)) in res.items():
        print(f'k32={k32} {cin}->{cout} @{H} pro={pro}: x3h max {mx3:.2e} rms {rms3:.2e} | f32 MFMA max {mx32:.2e} rms {rms32:.2e}')
        assert mx3 < 6e-7 and rms3 < 1.25 * rms32 + 1e-9
    # outputs + partials of the two kernels on one input
    n = 2
    x = (_rand((n, H, H, cin), 41) * 1.3 + 0.1).to(dev)
    w, b = _rand((cout, cin, 3, 3), 42, 0.08).to(dev), _rand((cout,), 43).to(dev)
    gamma = (_rand((cout,), 44) * 0.3 + 1).to(dev)
    wp = ops.pack_conv3_x3h(w)
    outs, stats = [], []
    try:
        for k32 in (1, 0):
            _lib.select(_lib.SEL_CONV_X3H_K32, k32)
            out = torch.empty((n * H * H, cout), device=dev)
            part = ops.new_gn_part(n, H, H, dev)
            part.fill_(float('nan'))
            ops.igemm(x, wp, n * H * H, cin, cout, out, bias=b, mode=ops.MODE_CONV3_S1, Hin=H, Win=H, Hout=H, Wout=H, x3h=True, gn_part=part)
            assert torch.isfinite(part).all()
            mean_f, scale_f = ops.groupnorm_finalize(part, gamma, n, H * H, cout)
            mean_s, scale_s = ops.groupnorm_stats(out, gamma, n, H * H, cout)
            assert (mean_f - mean_s).abs().max().item() < 2e-6 * (1 + mean_s.abs().max().item())
            assert ((scale_f - scale_s).abs() / scale_s.abs()).max().item() < 5e-6
            outs.append(out)
            stats.append((mean_f, scale_f))
    finally:
        _lib.select(_lib.SEL_CONV_X3H_K32, 1)
    assert not torch.equal(outs[0], outs[1])                                    # (two kernels really ran)
    assert (outs[0] - outs[1]).abs().max().item() < 2e-6 * outs[1].abs().max().item()
    assert (stats[0][0] - stats[1][0]).abs().max().item() < 1e-6 * (1 + stats[1][0].abs().max().item())


@pytest.mark.parametrize('k32', [1, 0])
def test_conv3_halo_x3h_groupnorm_prologue_without_swish_and_odd_batches(dev, k32):
    """the GroupNorm-apply prologue without the swish (Normalize alone), 5 images (an odd count of 16 x 16 maps: 10 tiles), on both MFMA shapes"""
    from viewformer_amd import _lib
    _lib.select(_lib.SEL_CONV_X3H_K32, k32)
    try:
        for cin, cout, H, n in ((128, 128, 16, 5), (64, 128, 32, 1)):
            (mx3, rms3), (mx32, rms32) = _run(dev, 's1', cin, cout, H, True, n=n, swish=False)
            assert mx3 < 6e-7 and rms3 < 1.25 * rms32 + 1e-9, (k32, cin, cout, H, mx3, rms3, rms32)
    finally:
        _lib.select(_lib.SEL_CONV_X3H_K32, 1)


@pytest.mark.parametrize('xscale,wscale', [(1e-2, 0.05), (300.0, 0.02), (1.0, 1e-4), (1.0, 30.0), (3e-3, 2.0)])
def test_x3h_holds_over_the_magnitudes_of_the_inference_path(dev, xscale, wscale):
    """un-normalised inputs (no prologue) from 1e-2 to a few hundred and any weight scale (absorbed by the pack-time power of two)"""
    (mx3, rms3), (mx32, rms32) = _run(dev, 's1', 128, 128, 16, False, xscale=xscale, wscale=wscale)
    print(f'x ~{xscale:g} w ~{wscale:g}: x3h max {mx3:.2e} rms {rms3:.2e} | f32 MFMA max {mx32:.2e} rms {rms32:.2e}')
    assert mx3 < 6e-7 and rms3 < 1.25 * rms32 + 1e-9


def test_x3h_range_limit_is_what_the_header_says(dev):
    """gradient-like magnitudes (1e-6) are below fp16's range: the error grows to ~1e-5 — the reason the backward pass keeps x6"""
    (mx3, rms3), (mx32, rms32) = _run(dev, 's1', 128, 128, 16, False, xscale=1e-6)
    print(f'x ~1e-6: x3h rms {rms3:.2e} | f32 MFMA rms {rms32:.2e}')
    assert rms3 > 3 * rms32 and rms3 < 1e-3


def test_x3h_fused_groupnorm_partials(dev):
    from viewformer_amd import ops
    for mode, cin, cout, H, n in [('s1', 64, 128, 16, 2), ('up', 32, 256, 8, 2), ('s1', 64, 512, 8, 3)]:
        x = (_rand((n, H, H, cin), 31) * 1.3 + 0.1).to(dev)
        w, b = _rand((cout, cin, 3, 3), 32, 0.08).to(dev), _rand((cout,), 33).to(dev)
        gamma = (_rand((cout,), 34) * 0.3 + 1).to(dev)
        m, Ho = (ops.MODE_CONV3_S1, H) if mode == 's1' else (ops.MODE_CONV3_UP2, 2 * H)
        out = torch.empty((n * Ho * Ho, cout), device=dev)
        part = ops.new_gn_part(n, Ho, Ho, dev)
        part.fill_(float('nan'))
        ops.igemm(x, ops.pack_conv3_x3h(w), n * Ho * Ho, cin, cout, out, bias=b, mode=m, Hin=H, Win=H, Hout=Ho, Wout=Ho, x3h=True,
                  gn_part=part)
        assert torch.isfinite(part).all()
        mean_f, scale_f = ops.groupnorm_finalize(part, gamma, n, Ho * Ho, cout)
        mean_s, scale_s = ops.groupnorm_stats(out, gamma, n, Ho * Ho, cout)
        assert (mean_f - mean_s).abs().max().item() < 2e-6 * (1 + mean_s.abs().max().item())
        assert ((scale_f - scale_s).abs() / scale_s.abs()).max().item() < 5e-6


def test_x3h_refuses_unsupported_shapes(dev):
    from viewformer_amd import ops
    assert ops.conv3_x3h_supported(ops.MODE_CONV3_S2PAD, 128, 128, 8, 8)           # 16x16 -> 8x8: the pair form (two images per tile)
    assert not ops.conv3_x3h_supported(ops.MODE_CONV3_S2PAD, 128, 128, 4, 4)       # smaller maps stay on the generic kernel
    x = torch.zeros((2 * 16 * 12, 64), device=dev)
    out = torch.empty((2 * 16 * 12, 128), device=dev)
    with pytest.raises(ops._lib.VfError):        # W % 16 != 0: refused, never rerouted
        ops.igemm(x, ops.pack_conv3_x3h(torch.zeros((128, 64, 3, 3), device=dev)), 2 * 16 * 12, 64, 128, out, mode=ops.MODE_CONV3_S1,
                  Hin=16, Win=12, Hout=16, Wout=12, x3h=True)


@pytest.mark.parametrize('M,K,N,epi,res', [(128, 64, 128, 0, False), (300, 128, 64, 0, True), (448, 768, 2304, 0, False),
                                           (1000, 3072, 768, 0, True), (70, 256, 1024, 1, True), (513, 768, 3072, 1, False)])
def test_gemm_x3h_is_fp32_equivalent(dev, M, K, N, epi, res):
    from viewformer_amd import ops
    x, w, b, r = _rand((M, K), 1), _rand((K, N), 2, 0.1), _rand((N,), 3), _rand((M, N), 4)
    pre = x.double() @ w.double() + b.double()
    ref = F.gelu(pre) if epi else pre
    mag = x.double().abs() @ w.double().abs() + b.double().abs()
    if res:
        ref, mag = ref + r.double(), mag + r.double().abs()
    kw = dict(bias=b.to(dev), res=r.to(dev) if res else None, epilogue=ops.EPI_GELU if epi else ops.EPI_NONE)
    o3, o32, o3t = (torch.empty((M, N), device=dev) for _ in range(3))
    ops.igemm(x.to(dev), ops.pack_dense_kn_x3h(w.to(dev)), M, K, N, o3, x3h=True, **kw)
    ops.igemm(x.to(dev), ops.pack_dense_nk_x3h(w.t().contiguous().to(dev)), M, K, N, o3t, x3h=True, **kw)
    ops.igemm(x.to(dev), ops.pack_dense_kn(w.to(dev)), M, K, N, o32, **kw)
    assert torch.equal(o3, o3t)                       # both packings describe the same matrix
    (mx3, rms3), (mx32, rms32) = _err(o3, ref, mag), _err(o32, ref, mag)
    print(f'gemm {M}x{K}x{N} epi={epi}: x3h max {mx3:.2e} rms {rms3:.2e} | f32 MFMA max {mx32:.2e} rms {rms32:.2e}')
    assert mx3 < 6e-7 and rms3 < 1.25 * rms32 + 1e-9
    with pytest.raises(ops._lib.VfError):             # K % 64 != 0 is refused, never rerouted
        ops.igemm(x[:, :32].contiguous().to(dev), ops.pack_dense_kn_x3h(w[:32].contiguous().to(dev)), M, 32, N, o3, x3h=True)


def test_migt_logits_x3h_vs_native_f32(dev):
    """full-size transformer: the x3h dense arm is as close to the fp64 oracle as the native f32-MFMA arm"""
    from oracle import migt_oracle as mg
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    cfg = MIGTConfig(sequence_size=4, localization_weight='1', pose_multiplier=0.2)
    sd = make_migt_weights(cfg, seed=0)
    g = np.random.Generator(np.random.PCG64(17))
    B, S = 2, 4
    codes = torch.from_numpy(g.integers(0, 1024, size=(B, S, 8, 8)))
    _, cams = synthetic_scene_batch(B, S, 8, 6)
    cams = mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(cams))[0])
    ids = torch.cat([codes[:, :-1], torch.full_like(codes[:, :1], 1024)], 1)
    ref = mg.migt_forward(sd, cfg, ids, cams, dtype=torch.float64)['logits'][:, -1]
    errs = {}
    for arith in ('f32', 'x3h'):
        m = MIGT(cfg, dense_arith=arith).load_state_dict(sd).to(dev)
        lg, _ = m.generate_and_localize(codes.to(dev), cams.to(dev))
        errs[arith] = ((lg.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    print(f'logit error vs fp64 (relative to max |logit|): native f32 {errs["f32"]:.2e}, x3h {errs["x3h"]:.2e}')
    assert errs['x3h'] < 1e-4 and errs['x3h'] < 2.0 * errs['f32'] + 1e-6


@pytest.mark.parametrize('u8', [True, False])
def test_conv_in_x3h_matches_fp64_and_the_valu_kernel(dev, u8):
    """encoder.conv_in on the matrix pipe: error vs fp64 no larger than the VALU fp32 kernel's; fused GroupNorm partials"""
    from viewformer_amd import ops
    n, H, W, C = 3, 32, 48, 128
    g = np.random.Generator(np.random.PCG64(5))
    img = torch.from_numpy(g.integers(0, 256, size=(n, H, W, 3), dtype=np.uint8))
    xf = (img.float() * torch.tensor(1.0 / 255)) * 2 - 1
    w, b = _rand((C, 3, 3, 3), 21, 0.2), _rand((C,), 22)
    ref = F.conv2d(xf.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    mag = F.conv2d(xf.double().abs().permute(0, 3, 1, 2), w.double().abs(), b.double().abs(), padding=1).permute(0, 2, 3, 1)
    src = img.to(dev) if u8 else xf.to(dev)
    part = ops.new_gn_part(n, H, W, dev)
    part.fill_(float('nan'))
    o3 = ops.conv_in(src, w.to(dev), b.to(dev), n, H, W, C, wp3h=ops.pack_conv_in_x3h(w.to(dev)), gn_part=part)
    ov = ops.conv_in(src, w.to(dev), b.to(dev), n, H, W, C)
    (mx3, rms3), (mxv, rmsv) = _err(o3, ref, mag), _err(ov, ref, mag)
    print(f'conv_in u8={u8}: x3h max {mx3:.2e} rms {rms3:.2e} | VALU fp32 max {mxv:.2e} rms {rmsv:.2e}')
    assert mx3 < 4e-7 and rms3 < 1.25 * rmsv + 1e-9
    gamma = (_rand((C,), 23) * 0.3 + 1).to(dev)
    mean_f, scale_f = ops.groupnorm_finalize(part, gamma, n, H * W, C)
    mean_s, scale_s = ops.groupnorm_stats(o3.view(n * H * W, C), gamma, n, H * W, C)
    assert (mean_f - mean_s).abs().max().item() < 2e-6 * (1 + mean_s.abs().max().item())
    assert ((scale_f - scale_s).abs() / scale_s.abs()).max().item() < 5e-6
    assert not ops.conv_in_x3h_supported(20, 48, 128)
