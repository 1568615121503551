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
                                                  ('s1', 128, 128, 8, True), ('s1', 512, 512, 8, False), ('s1', 256, 256, 16, True),
                                                  # Downsample: pad (right, bottom) + stride 2
                                                  ('s2', 128, 128, 32, False), ('s2', 64, 256, 64, False), ('s2', 32, 128, 32, False)])
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
    m, Ho = {'s1': (ops.MODE_CONV3_S1, H), 'up': (ops.MODE_CONV3_UP2, 2 * H), 's2': (ops.MODE_CONV3_S2PAD, H // 2)}[mode]
    if mode == 'up':
        a = F.interpolate(a, scale_factor=2.0, mode='nearest')
    if mode == 's2':
        ap = F.pad(a, (0, 1, 0, 1))
        ref = F.conv2d(ap, w.double(), None, stride=2)
        mag = F.conv2d(ap.abs(), w.double().abs(), None, stride=2)
    else:
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
    assert not ops.conv3_x6_supported(ops.MODE_CONV3_S2PAD, 128, 128, 8, 8)       # 16x16 -> 8x8 stays on the generic kernel
    assert not ops.conv3_x6_supported(ops.MODE_CONV3_S1, 128, 64, 64, 64)


@pytest.mark.parametrize('kernel', ['x6', 'bf16'])
@pytest.mark.parametrize('mode,cin,cout,H,n', [('s1', 64, 128, 16, 2), ('up', 32, 256, 8, 2), ('s1', 64, 512, 8, 3), ('s1', 32, 128, 8, 4),
                                                ('s1', 32, 1024, 32, 1)])
def test_fused_groupnorm_partials_match_the_standalone_statistics(dev, kernel, mode, cin, cout, H, n):
    """the conv epilogue's partial {sum, sumsq} -> finalize == groupnorm_stats of the stored output (pair tiles with an odd
    image count included: the duplicated half must not be counted)"""
    from viewformer_amd import ops
    x = (_rand((n, H, H, cin), 31) * 1.3 + 0.1).to(dev)
    w, b = _rand((cout, cin, 3, 3), 32, 0.08).to(dev), _rand((cout,), 33).to(dev)
    gamma = (_rand((cout,), 34) * 0.3 + 1).to(dev)
    m, Ho = (ops.MODE_CONV3_S1, H) if mode == 's1' else (ops.MODE_CONV3_UP2, 2 * H)
    res = _rand((n * Ho * Ho, cout), 35).to(dev)
    out = torch.empty((n * Ho * Ho, cout), device=dev)
    part = ops.new_gn_part(n, Ho, Ho, dev)
    part.fill_(float('nan'))                              # every slot must be written
    wp = ops.pack_conv3_x6(w) if kernel == 'x6' else ops.pack_conv3_bf16(w)
    ops.igemm(x, wp, n * Ho * Ho, cin, cout, out, bias=b, res=res, mode=m, Hin=H, Win=H, Hout=Ho, Wout=Ho,
              x6=kernel == 'x6', bf16=kernel == 'bf16', gn_part=part)
    assert torch.isfinite(part).all()
    mean_f, scale_f = ops.groupnorm_finalize(part, gamma, n, Ho * Ho, cout)
    mean_s, scale_s = ops.groupnorm_stats(out, gamma, n, Ho * Ho, cout)
    assert (mean_f - mean_s).abs().max().item() < 2e-6 * (1 + mean_s.abs().max().item())
    assert ((scale_f - scale_s).abs() / scale_s.abs()).max().item() < 5e-6
    # and the stored output is the same with and without the statistics
    out2 = torch.empty_like(out)
    ops.igemm(x, wp, n * Ho * Ho, cin, cout, out2, bias=b, res=res, mode=m, Hin=H, Win=H, Hout=Ho, Wout=Ho,
              x6=kernel == 'x6', bf16=kernel == 'bf16')
    assert torch.equal(out, out2)


def test_fused_statistics_are_refused_by_the_generic_kernels(dev):
    from viewformer_amd import ops
    x = torch.zeros((256, 64), device=dev)
    out = torch.empty((256, 128), device=dev)
    part = torch.zeros((1, 2, 32, 2), device=dev)
    with pytest.raises(ops._lib.VfError):
        ops.igemm(x, ops.pack_dense_kn(torch.zeros((64, 128), device=dev)), 256, 64, 128, out, gn_part=part)


@pytest.mark.parametrize('M,K,N,epi,res', [(128, 64, 128, 0, False), (300, 128, 64, 0, True), (448, 768, 2304, 0, False),
                                           (1000, 3072, 768, 0, True), (70, 256, 1024, 1, True), (513, 768, 3072, 1, False)])
def test_gemm_x6_is_fp32_equivalent(dev, M, K, N, epi, res):
    from viewformer_amd import ops
    x, w, b, r = _rand((M, K), 1), _rand((K, N), 2, 0.1), _rand((N,), 3), _rand((M, N), 4)
    pre = x.double() @ w.double() + b.double()
    ref = F.gelu(pre) if epi else pre
    mag = x.double().abs() @ w.double().abs() + b.double().abs()
    if res:
        ref, mag = ref + r.double(), mag + r.double().abs()
    kw = dict(bias=b.to(dev), res=r.to(dev) if res else None, epilogue=ops.EPI_GELU if epi else ops.EPI_NONE)
    o6, o32, o6t = (torch.empty((M, N), device=dev) for _ in range(3))
    ops.igemm(x.to(dev), ops.pack_dense_kn_x6(w.to(dev)), M, K, N, o6, x6=True, **kw)
    ops.igemm(x.to(dev), ops.pack_dense_nk_x6(w.t().contiguous().to(dev)), M, K, N, o6t, x6=True, **kw)
    ops.igemm(x.to(dev), ops.pack_dense_kn(w.to(dev)), M, K, N, o32, **kw)
    assert torch.equal(o6, o6t)                       # both packings describe the same matrix
    (mx6, rms6), (mx32, rms32) = _err(o6, ref, mag), _err(o32, ref, mag)
    print(f'gemm {M}x{K}x{N} epi={epi}: x6 max {mx6:.2e} rms {rms6:.2e} | f32 MFMA max {mx32:.2e} rms {rms32:.2e}')
    assert mx6 < 6e-7 and rms6 < 1.25 * rms32 + 1e-9
    with pytest.raises(ops._lib.VfError):             # K % 64 != 0 is refused, never rerouted
        ops.igemm(x[:, :32].contiguous().to(dev), ops.pack_dense_kn_x6(w[:32].contiguous().to(dev)), M, 32, N, o6, x6=True)


def test_gemm_x6_groupnorm_prologue(dev):
    """1x1 convolution with the fused GroupNorm-apply (AttnBlock q/k/v: no swish; and with swish)"""
    from viewformer_amd import ops
    n, HW, C, N = 3, 64, 256, 768
    x = (_rand((n * HW, C), 41) * 1.5 + 0.3).to(dev)
    gamma, beta = (_rand((C,), 42) * 0.3 + 1).to(dev), (_rand((C,), 43) * 0.2).to(dev)
    w, b = _rand((C, N), 44, 0.05).to(dev), _rand((N,), 45).to(dev)
    mean_c, scale_c = ops.groupnorm_stats(x, gamma, n, HW, C)
    for swish in (False, True):
        o6, o32 = torch.empty((n * HW, N), device=dev), torch.empty((n * HW, N), device=dev)
        kw = dict(bias=b, pro=(mean_c, scale_c, beta), pro_swish=swish, pro_rows_per_img=HW)
        ops.igemm(x, ops.pack_dense_kn_x6(w), n * HW, C, N, o6, x6=True, **kw)
        ops.igemm(x, ops.pack_dense_kn(w), n * HW, C, N, o32, **kw)
        assert (o6 - o32).abs().max().item() < 2e-5 * o32.abs().max().item()


def test_migt_logits_x6_vs_native_f32(dev):
    """full-size transformer: the x6 dense arm is as close to the fp64 oracle as the native f32-MFMA arm"""
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
    for arith in ('f32', 'x6'):
        m = MIGT(cfg, dense_arith=arith).load_state_dict(sd).to(dev)
        lg, _ = m.generate_and_localize(codes.to(dev), cams.to(dev))
        errs[arith] = ((lg.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    print(f'logit error vs fp64 (relative to max |logit|): native f32 {errs["f32"]:.2e}, x6 {errs["x6"]:.2e}')
    assert errs['x6'] < 1e-4 and errs['x6'] < 2.0 * errs['f32'] + 1e-6


def _attn_oracle(qkv, B, H, T, L, d):
    from oracle import migt_oracle as mg
    x = qkv.double().view(B, T // L, L, 3 * d)
    v, q, k = x.chunk(3, -1)
    sp = lambda t: mg._split_heads(t, H)
    return mg._merge_heads(mg.compute_causal_block_attention(sp(k), sp(v), sp(q))).reshape(B * T, d)


@pytest.mark.parametrize('B,H,S,L,scale', [(2, 2, 4, 16, 0.35), (1, 12, 7, 64, 0.35), (1, 2, 5, 48, 0.35), (1, 2, 4, 64, 2.0)])
def test_attention_x6_is_fp32_equivalent(dev, B, H, S, L, scale):
    """x6 attention vs the fp64 oracle: error no larger than the native f32-MFMA kernel's (incl. |q.k| in the hundreds)"""
    from viewformer_amd import ops
    d, T = H * 64, S * L
    qkv = _rand((B * T, 3 * d), 71, scale)
    ref = _attn_oracle(qkv, B, H, T, L, d)
    g = qkv.to(dev)
    o6, o32, o6d = (torch.empty((B * T, d), device=dev) for _ in range(3))
    a = (g[:, d:2 * d], g[:, 2 * d:], g[:, :d])
    ops.attn_blockcausal(*a, o6, B, H, T, L, 3 * d, 3 * d, 3 * d, d, 1.0, True, x6=True)
    ops.attn_blockcausal(*a, o6d, B, H, T, L, 3 * d, 3 * d, 3 * d, d, 1.0, False, x6=True)
    ops.attn_blockcausal(*a, o32, B, H, T, L, 3 * d, 3 * d, 3 * d, d, 1.0, True)
    assert torch.equal(o6, o6d)                          # skipping masked tiles == the dense "-1e4" form
    e6 = (o6.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    e32 = (o32.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f'attention B={B} H={H} S={S} L={L} scale={scale}: x6 {e6:.2e} | native f32 {e32:.2e}')
    assert e6 < 1e-4 and e6 < 1.5 * e32 + 1e-7


@pytest.mark.parametrize('mode', ['twin', 'streams'])
def test_attention_x6_masks(dev, mode):
    from viewformer_amd import ops
    B, H, S, L = 2, 3, 4, 64
    d = H * 64
    NS = 3 if mode == 'streams' else 1
    T = NS * S * L
    spec = S - 2 if mode == 'twin' else -S
    qkv = _rand((B * T, 3 * d), 5, 0.35).to(dev)
    a = (qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d])
    o6, o32 = torch.empty((B * T, d), device=dev), torch.empty((B * T, d), device=dev)
    ops.attn_blockcausal(*a, o6, B, H, T, L, 3 * d, 3 * d, 3 * d, d, 1.0, True, spec, x6=True)
    ops.attn_blockcausal(*a, o32, B, H, T, L, 3 * d, 3 * d, 3 * d, d, 1.0, True, spec)
    assert (o6 - o32).abs().max().item() < 2e-5 * o32.abs().max().item()


@pytest.mark.parametrize('M,K,N,S', [(768, 4096, 768, 4), (256, 1920, 384, 7), (128, 64, 128, 3)])
def test_gemm_x6_split_k(dev, M, K, N, S):
    """split-K (the training step's dW GEMMs): slabs + ordered sum == the unsplit GEMM to fp32 rounding, accumulate semantics"""
    from viewformer_amd import ops
    x, w = _rand((M, K), 1), _rand((K, N), 2, 0.1)
    wp = ops.pack_dense_kn_x6(w.to(dev))
    base = _rand((M, N), 3).to(dev)
    dst = base.clone()
    ops.gemm_x6_splitk(x.to(dev), wp, M, K, N, dst, S)
    ref = base.double().cpu() + x.double() @ w.double()
    mag = base.double().cpu().abs() + x.double().abs() @ w.double().abs()
    assert ((dst.double().cpu() - ref).abs() / mag).max().item() < 6e-7
    dst2 = torch.full((M, N), float('nan'), device=dev)
    ops.gemm_x6_splitk(x.to(dev), wp, M, K, N, dst2, S, accumulate=False)
    one = torch.empty((M, N), device=dev)
    ops.igemm(x.to(dev), wp, M, K, N, one, x6=True)
    assert (dst2 - one).abs().max().item() < 1e-5 * one.abs().max().item()
    ops.gemm_x6_splitk(x.to(dev), wp, M, K, N, dst2, S, accumulate=False)          # deterministic
    d3 = dst2.clone()
    ops.gemm_x6_splitk(x.to(dev), wp, M, K, N, dst2, S, accumulate=False)
    assert torch.equal(d3, dst2)
