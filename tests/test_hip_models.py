"""GPU: the model objects (VQGAN / MIGT / generate_batch_predictions) against the golden vectors
recorded from the reference and against the CPU oracle, through the C-ABI."""
import numpy as np
import pytest
import torch

from conftest import TINY_MIGT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs the MI355X'
    return torch.device('cuda:0')


def _vq_model(cfg, sd, dev, fmt='NCHW', arith='x6'):
    from viewformer_amd.vqgan import VQGAN
    return VQGAN(cfg, data_format=fmt, conv_arith=arith).load_state_dict(sd).to(dev)


def _maxerr(a, b):
    return (a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max().item()


# ------------------------------------------------------------------------------------------------ VQGAN
@pytest.mark.parametrize('arith', ['f32', 'x6', 'x3h'])
def test_vqgan_tiny_matches_reference_golden(dev, tiny_vq, arith):
    from oracle import vqgan_oracle as vq
    cfg, sd, g = tiny_vq
    m = _vq_model(cfg, sd, dev, arith=arith)
    x = vq.preprocess_u8(torch.from_numpy(g['frames']))
    quant, diff, codes = m.encode(x.to(dev))
    assert codes.dtype == torch.int64 and tuple(codes.shape) == g['codes'].shape
    bad = codes.cpu().numpy() != g['codes']
    assert bad.sum() == 0, f'{bad.sum()} code mismatches; margins {g["margin"][bad.reshape(-1)]}'
    assert _maxerr(quant, g['quant']) < 1e-5
    assert abs(float(diff) - float(g['diff'])) < 1e-6
    dec = m.decode_code(torch.from_numpy(g['codes']).to(dev))
    assert _maxerr(dec, g['decoded']) < 2e-5                  # utils/testing.py tolerance class (1e-5)
    # uint8 NHWC entry == float entry (the fused TF-style preprocess is bit-identical)
    codes_u8 = m.encode(torch.from_numpy(g['frames']).to(dev))[-1]
    assert torch.equal(codes_u8, codes)


@pytest.mark.parametrize('arith', ['f32', 'x6', 'x3h'])      # native f32 MFMA / fp32-equivalent split-bf16 / split-fp16 convolutions
def test_vqgan_full_matches_reference_golden(dev, full_vq, arith):
    from viewformer_amd.weights import synthetic_scene_batch
    cfg, sd, g = full_vq
    m = _vq_model(cfg, sd, dev, 'NHWC', arith=arith)
    frames, _ = synthetic_scene_batch(1, 4, 128, seed=int(g['input_seed']))
    res = m.encode(torch.from_numpy(frames[0]).to(dev))
    codes = res[-1].cpu().numpy()
    bad = codes != g['codes']
    assert bad.sum() == 0, f'{bad.sum()} / {bad.size} code mismatches; margins {g["margin"][bad.reshape(-1)]}'
    z = res._z.view(4, 8, 8, 256).permute(0, 3, 1, 2)
    dec = m.decode_code(torch.from_numpy(g['codes'][:2]).to(dev))            # NHWC out
    assert tuple(dec.shape) == (2, 128, 128, 3)
    ez, ed = _maxerr(z, g['z']), _maxerr(dec.permute(0, 3, 1, 2), g['decoded'])
    print(f'conv_arith={arith}: max |z - ref| {ez:.2e}, max |decoded - ref| {ed:.2e}')
    assert ez < 5e-5 and ed < 5e-5


def test_vqgan_batch_and_chunk_invariance(dev, full_vq):
    """size-independent property: an image's codes / pixels do not depend on its batch-mates
    (bit-exact), nor on max_images_per_call chunking."""
    from viewformer_amd.weights import synthetic_scene_batch
    cfg, sd, g = full_vq
    m = _vq_model(cfg, sd, dev, 'NHWC')
    frames, _ = synthetic_scene_batch(2, 5, 128, seed=9)
    imgs = torch.from_numpy(frames.reshape(10, 128, 128, 3)).to(dev)
    all_codes = m.encode(imgs)[-1]
    m.max_images_per_call = 3
    assert torch.equal(m.encode(imgs)[-1], all_codes)
    assert torch.equal(m.encode(imgs[4:7])[-1], all_codes[4:7])
    m.max_images_per_call = 256
    dec = m.decode_code(all_codes)
    assert torch.equal(m.decode_code(all_codes[7:9]), dec[7:9])
    assert len(torch.unique(all_codes)) > 30


def test_vqgan_load_state_dict_contract(dev, tiny_vq):
    from viewformer_amd.vqgan import VQGAN
    cfg, sd, _ = tiny_vq
    m = VQGAN(cfg)
    extra = dict(sd)
    extra['perceptual_loss.net.weight'] = np.zeros(3, np.float32)         # ignored like vqgan_th.py:322
    m.load_state_dict(extra)
    missing = dict(sd)
    del missing['quant_conv.bias']
    with pytest.raises(RuntimeError, match='Missing keys'):
        VQGAN(cfg).load_state_dict(missing)
    bad = dict(sd)
    bad['encoder.bogus'] = np.zeros(1, np.float32)
    with pytest.raises(RuntimeError, match='Unexpected keys'):
        VQGAN(cfg).load_state_dict(bad)


# ------------------------------------------------------------------------------------------------ MIGT
def _migt(cfg, sd, dev):
    from viewformer_amd.migt import MIGT
    return MIGT(cfg).load_state_dict(sd).to(dev)


@pytest.mark.parametrize('loc', [False, True])
def test_migt_tiny_matches_oracle(dev, loc):
    from oracle import migt_oracle as mg
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    cfg = MIGTConfig(**TINY_MIGT, localization_weight='1' if loc else '0', pose_multiplier=0.2)
    sd = make_migt_weights(cfg, seed=2, std=0.08)
    g = np.random.Generator(np.random.PCG64(13))
    B, S, t = 3, 4, cfg.token_image_size
    ids = torch.from_numpy(g.integers(0, cfg.n_embeddings + 1, size=(B, S, t, t)))
    _, cams = synthetic_scene_batch(B, S, 8, 4)
    cams = mg.normalize_cameras(torch.from_numpy(cams))
    m = _migt(cfg, sd, dev)
    poses = cams[:, :-1] if loc else cams
    out = m(dict(input_ids=ids.to(dev), poses=poses.to(dev)))
    ref = mg.migt_forward(sd, cfg, ids, poses, dtype=torch.float64)
    assert tuple(out['logits'].shape) == (B, S, t, t, cfg.n_embeddings)
    assert _maxerr(out['logits'], ref['logits']) < 2e-4
    assert _maxerr(out['hidden_states'][0], ref['hidden_states'][0]) < 2e-4
    if loc:
        assert _maxerr(out['pose_prediction'], ref['pose_prediction']) < 2e-4
        rc = m.reduce_cameras(out['pose_prediction'][:, -1:], -2)
        assert _maxerr(rc, mg.reduce_cameras(ref['pose_prediction'][:, -1:], -2)) < 2e-4


def test_migt_full_config_matches_oracle(dev):
    """BASELINE config #2 shape: 12 layers, d=768, 12 heads, 6 context views + MASK view."""
    from oracle import migt_oracle as mg
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    cfg = MIGTConfig(sequence_size=7, localization_weight='0', pose_multiplier=0.2)
    sd = make_migt_weights(cfg, seed=0)
    g = np.random.Generator(np.random.PCG64(17))
    B, S = 2, 7
    ids = torch.from_numpy(g.integers(0, 1024, size=(B, S, 8, 8)))
    ids[:, -1] = cfg.n_embeddings                       # MASK view
    _, cams = synthetic_scene_batch(B, S, 8, 6)
    cams = mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(cams))[0])
    m = _migt(cfg, sd, dev)
    out = m(dict(input_ids=ids.to(dev), poses=cams.to(dev)))
    ref64 = mg.migt_forward(sd, cfg, ids, cams, dtype=torch.float64)['logits']
    ref32 = mg.migt_forward(sd, cfg, ids, cams, dtype=torch.float32)['logits']
    e_hip = _maxerr(out['logits'], ref64)
    e_cpu = _maxerr(ref32, ref64)
    print(f'logit err vs fp64: hip {e_hip:.3e}, torch-cpu fp32 {e_cpu:.3e}; |logit| max {ref64.abs().max():.3f}')
    assert e_hip < 1e-3                                  # stated fp32 tolerance for logits
    assert e_hip < 20 * e_cpu + 1e-5                     # same error class as the fp32 reference arm
    last = m(dict(input_ids=ids.to(dev), poses=cams.to(dev)), last_view_logits_only=True)['logits_last']
    assert torch.equal(last, out['logits'][:, -1])
    agree = (out['logits'][:, -1].argmax(-1).cpu() == ref64[:, -1].argmax(-1)).float().mean().item()
    assert agree > 0.98, agree


def test_migt_matches_hugging_face_gpt2_golden(dev):
    """the HIP transformer against outputs of a THIRD-PARTY implementation: ``transformers``' GPT-2 run on MIGT weights
    (tests/golden/migt_hf_gpt2.npz, written by tests/golden/make_hf_gpt2_golden.py: no score scaling, (V, Q, K) split, block-causal 0 / -1e4
    mask, pose MLPs on transformers' Conv1D; the multi-context pass as one GPT-2 call per view).  Tiny shape: logits of every view on the
    fp32-equivalent arm, the multi-context pass's MASK-stream logits and LOC-stream camera predictions; full size (12 layers, d = 768):
    last-view logits on the fp32-equivalent arm (stated fp32 tolerance) and on the bf16 arm the bench times (stated bf16 tolerance)."""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.weights import make_migt_weights
    from conftest import load_golden
    g = load_golden('migt_hf_gpt2.npz')
    cfg = MIGTConfig(n_embeddings=64, n_head=2, d_model=128, n_layer=2, token_image_size=4, sequence_size=4, localization_weight='1', pose_multiplier=0.2)
    sd = make_migt_weights(cfg, seed=int(g['tiny_seed']), std=float(g['tiny_std']))
    ids, cams = torch.from_numpy(g['tiny_ids']).to(dev), torch.from_numpy(g['tiny_cams']).to(dev)
    m = MIGT(cfg).load_state_dict(sd).to(dev)
    out = m(dict(input_ids=ids, poses=cams))
    e1 = _maxerr(out['logits'], torch.from_numpy(g['tiny_logits']))
    S = ids.shape[1]
    in_ids = torch.cat([ids[:, :-1], torch.full_like(ids[:, :1], cfg.n_embeddings)], 1)
    ctx = torch.cat([cams[:, :-1], torch.zeros_like(cams[:, :1])], 1)
    multi = m(dict(input_ids=in_ids, poses=ctx, localization_tokens=ids[:, -1:].expand(-1, S, -1, -1).contiguous(),
                   output_poses=cams[:, -1:].expand(-1, S, -1).contiguous()), training=False)
    e2 = _maxerr(multi['logits'], torch.from_numpy(g['tiny_multi_logits']))
    e3 = _maxerr(multi['pose_prediction'], torch.from_numpy(g['tiny_multi_pose']))
    print(f'HIP vs HF GPT-2 golden (tiny): logits {e1:.2e}, multi-context logits {e2:.2e}, camera predictions {e3:.2e}')
    assert e1 < 2e-4 and e2 < 2e-4 and e3 < 2e-4
    fcfg = MIGTConfig(sequence_size=6, n_loss_skip=1, pose_multiplier=0.2, localization_weight='cosine(0,1,120000)')
    fsd = make_migt_weights(fcfg, seed=int(g['full_seed']))
    ref = torch.from_numpy(g['full_logits_last'])
    fids, fcams = torch.from_numpy(g['full_ids']).to(dev), torch.from_numpy(g['full_cams']).to(dev)
    for arm, tol in (('f32', 1e-3), ('bf16', 3e-2 * float(ref.abs().max()))):
        mm = MIGT(fcfg, precision=arm).load_state_dict(fsd).to(dev)
        last = mm(dict(input_ids=fids, poses=fcams), last_view_logits_only=True)['logits_last']
        err = _maxerr(last.reshape(ref.shape), ref)
        print(f'HIP vs HF GPT-2 golden (full size, {arm} arm): {err:.2e} of |logit| max {float(ref.abs().max()):.3f}')
        assert err < tol, (arm, err)
        # the localization pass (evaluate_transformer.py:134-140): LOC embedding on the last view, camera head on its tokens
        loc = mm(dict(input_ids=torch.from_numpy(g['full_loc_codes']).to(dev), poses=fcams[:, :-1].contiguous()))['pose_prediction'][:, -1]
        e_loc = _maxerr(loc, torch.from_numpy(g['full_loc_pose_last']))
        print(f'   localization pass, camera head: {e_loc:.2e}')
        assert e_loc < (2e-4 if arm == 'f32' else 5e-2), (arm, e_loc)


# ------------------------------------------------------------------------------------------------ pipeline
@pytest.mark.parametrize('B,S', [(2, 3), (1, 2), (3, 5)])          # (1, 2): a single context view; odd image counts hit the pair tiles' tail
def test_generate_batch_predictions_matches_oracle(dev, full_vq, B, S):
    from oracle import pipeline_oracle as po
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.evaluate import generate_batch_predictions
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    vcfg, vsd, _ = full_vq
    mcfg = MIGTConfig(sequence_size=S, localization_weight='1', pose_multiplier=0.2, n_layer=4)
    msd = make_migt_weights(mcfg, seed=1, std=0.05)
    frames, cams = synthetic_scene_batch(B, S, 128, seed=3)
    ref = po.generate_batch_predictions(msd, mcfg, vsd, vcfg, frames, cams, return_intermediates=True)
    vq_m = _vq_model(vcfg, vsd, dev, 'NHWC')
    tr_m = _migt(mcfg, msd, dev)
    got = generate_batch_predictions(tr_m, vq_m, frames, cams, return_codes=True)
    assert got['generated_images'].dtype == torch.uint8 and tuple(got['generated_images'].shape) == (B, 128, 128, 3)
    assert torch.equal(got['codes'].cpu(), ref['codes'])                         # context tokens bit-exact
    assert _maxerr(got['logits_last'], ref['logits_last']) < 1e-3
    same = got['generated_codes'].cpu() == ref['generated_codes']
    print('generated-code agreement', same.float().mean().item())
    assert same.float().mean() > 0.97
    if bool(same.all()):
        diff = (got['generated_images'].cpu().int() - ref['generated_images'].int()).abs()
        assert diff.max() <= 1 and (diff > 0).float().mean() < 0.01         # truncation flips only
    assert _maxerr(got['generated_cameras'], ref['generated_cameras']) < 1e-3
    assert torch.equal(got['ground_truth_images'].cpu(), torch.from_numpy(frames[:, -1]))


def test_no_cpu_fallback(dev, tiny_vq):
    from viewformer_amd import ops, _lib
    from viewformer_amd.vqgan import VQGAN
    cfg, sd, _ = tiny_vq
    with pytest.raises(_lib.VfError):
        VQGAN(cfg).load_state_dict(sd).to('cpu')
    with pytest.raises(_lib.VfError):
        ops.layernorm(torch.zeros(4, 128), torch.ones(128), torch.zeros(128), 4, 128)


def test_fused_generate_and_localize_is_bit_identical_to_two_passes(dev, full_vq):
    """the twin-view single pass == the reference's two separate transformer calls, bit for bit"""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.evaluate import generate_batch_predictions
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    vcfg, vsd, _ = full_vq
    mcfg = MIGTConfig(sequence_size=6, localization_weight='cosine(0,1,120000)', pose_multiplier=0.2, n_layer=3)
    msd = make_migt_weights(mcfg, seed=4, std=0.05)
    frames, cams = synthetic_scene_batch(2, 4, 128, seed=8)
    vq_m = _vq_model(vcfg, vsd, dev, 'NHWC')
    for skip in (True, False):
        tr_m = _migt(mcfg, msd, dev)
        tr_m.skip_masked = skip
        a = generate_batch_predictions(tr_m, vq_m, frames, cams, return_codes=True, fused_passes=True)
        b = generate_batch_predictions(tr_m, vq_m, frames, cams, return_codes=True, fused_passes=False)
        assert torch.equal(a['logits_last'], b['logits_last'])
        assert torch.equal(a['pose_last'], b['pose_last'])
        assert torch.equal(a['generated_images'], b['generated_images'])
        assert torch.equal(a['generated_cameras'], b['generated_cameras'])


@pytest.mark.parametrize('precision,S,B', [('f32', 4, 2), ('bf16', 7, 5), ('bf16', 2, 3)])
def test_last_block_on_the_two_returned_views_only_keeps_their_bits(dev, precision, S, B):
    """MIGT.prune_last_block (round 6): generate_and_localize runs the LAST block's projection, LayerNorm, MLP and ln_f on the MASK and LOC views' rows
    only (the other rows of the last block feed nothing).  Codes / logits and the pose prediction equal the full-rows pass bit for bit, on the fp32
    arm and on the bf16 arm (256-tile GEMMs at the reduced row count), for a two-view sequence as well."""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.weights import make_migt_weights
    mcfg = MIGTConfig(sequence_size=S, localization_weight='1', pose_multiplier=0.2, n_layer=2)
    msd = make_migt_weights(mcfg, seed=6, std=0.05)
    g = np.random.Generator(np.random.PCG64(S * 10 + B))
    codes = torch.from_numpy(g.integers(0, 1024, size=(B, S, 8, 8)))
    cams = torch.from_numpy(g.standard_normal((B, S, 7)).astype(np.float32))
    m = MIGT(mcfg, precision=precision).load_state_dict(msd).to(dev)
    outs = []
    for prune in (False, True):
        m.prune_last_block = prune
        for codes_only in (False, True):
            first, pose = m.generate_and_localize(codes, cams, codes_only=codes_only)
            outs.append((prune, codes_only, first.clone(), pose.clone()))
    torch.cuda.synchronize()
    for (p0, c0, f0, q0), (p1, c1, f1, q1) in zip(outs[:2], outs[2:]):
        assert c0 == c1 and not p0 and p1
        assert torch.equal(f0, f1) and torch.equal(q0.view(torch.int32), q1.view(torch.int32))


def test_generate_codes_dataset_with_the_gpu_codebook(dev, tiny_vq, tmp_path):
    """generate-codes end to end: frames -> VQGAN.encode on the GPU -> TFRecord code dataset -> read back"""
    from viewformer_amd import codes_dataset as cd
    cfg, sd, g = tiny_vq
    m = _vq_model(cfg, sd, dev, 'NHWC')
    frames = g['frames']                                    # [N,H,W,3] uint8
    n = frames.shape[0] // 2 * 2
    seqs = [dict(frames=frames[:n // 2], cameras=np.arange(n // 2 * 7, dtype=np.float32).reshape(-1, 7)),
            dict(frames=frames[n // 2:n], cameras=-np.arange(n // 2 * 7, dtype=np.float32).reshape(-1, 7))]
    out = str(tmp_path / 'codes' / 'tiny')
    info = cd.generate_codes(seqs, out, m, split='test', max_sequences_per_shard=1, batch_size=3)
    assert info['token_image_size'] == g['codes'].shape[-1] and info['test_size'] == 2
    back = list(cd.read_code_dataset(str(tmp_path / 'codes'), 'test'))
    got = np.concatenate([b['codes'] for b in back], 0)
    assert np.array_equal(got, g['codes'][:n])              # = the reference's own codes for these frames
    assert np.array_equal(back[1]['cameras'], seqs[1]['cameras'])


def test_load_model_from_a_reference_written_directory(dev, tiny_vq):
    """load_model on tests/golden/vqgan_tiny_model — config.json + Lightning-style model.ckpt written by the
    reference's own classes — reproduces the reference's codes for the golden frames"""
    import os
    from viewformer_amd.checkpoint import load_model
    _, _, g = tiny_vq
    here = os.path.dirname(os.path.abspath(__file__))
    m = load_model(os.path.join(here, 'golden', 'vqgan_tiny_model', 'model.ckpt'), device=dev)
    assert m.config.image_size == 32 and m.config.n_embed == 64 and m.config.stride == 2
    codes = m.encode(torch.from_numpy(g['frames']).to(dev))[-1]          # TF-convention entry (NHWC uint8), as the evaluators use
    assert np.array_equal(codes.cpu().numpy(), g['codes'])
    dec = m.decode_code(torch.from_numpy(g['codes']).to(dev)).permute(0, 3, 1, 2)
    assert _maxerr(dec, g['decoded']) < 2e-5


def test_full_size_pipeline_properties(dev, full_vq):
    """BASELINE configs[1] shapes (full VQGAN + 12-layer MIGT, 7 views): size-independent properties instead of the (too slow)
    oracle — run-to-run determinism, scene independence of every output (a scene alone == the same scene inside a batch, bit for
    bit), and context tokens identical across the fp32 and mixed arms"""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.evaluate import generate_batch_predictions
    from viewformer_amd.migt import MIGT
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    vcfg, vsd, _ = full_vq
    mcfg = MIGTConfig(sequence_size=7, localization_weight='1', pose_multiplier=0.2)
    msd = make_migt_weights(mcfg, seed=0)
    frames, cams = synthetic_scene_batch(5, 7, 128, seed=11)
    outs = {}
    for arm in ('f32', 'bf16'):
        vq_m = VQGAN(vcfg, data_format='NHWC', decoder_precision=arm).load_state_dict(vsd).to(dev)
        tr_m = MIGT(mcfg, precision=arm).load_state_dict(msd).to(dev)
        a = generate_batch_predictions(tr_m, vq_m, frames, cams, return_codes=True)
        b = generate_batch_predictions(tr_m, vq_m, frames, cams, return_codes=True)
        for key in ('codes', 'generated_codes', 'generated_images', 'generated_cameras'):
            assert torch.equal(a[key], b[key]), (arm, key)                                   # deterministic
        one = generate_batch_predictions(tr_m, vq_m, frames[3:4], cams[3:4], return_codes=True)
        for key in ('codes', 'generated_codes', 'generated_images', 'generated_cameras'):
            assert torch.equal(one[key][0], a[key][3]), (arm, key)                           # scene independence
        outs[arm] = a
    assert torch.equal(outs['f32']['codes'], outs['bf16']['codes'])                          # the encoder is fp32 in both arms
    assert outs['f32']['generated_images'].dtype == torch.uint8


def test_quantizer_ema_training_matches_reference_golden(dev):
    """QuantizeEMA.forward training branch on the GPU (vq_train.QuantizeEMATrainer) vs three steps recorded from the reference"""
    import os
    from viewformer_amd.vq_train import QuantizeEMATrainer
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'vq_ema.npz'))
    tr = QuantizeEMATrainer(torch.from_numpy(g['E0']).to(dev), decay=float(g['decay']), eps=float(g['eps']))
    for step in range(3):
        q, diff, ind = tr(torch.from_numpy(g[f'z{step}']).to(dev))
        assert np.array_equal(ind.cpu().numpy(), g[f'ind{step}'])
        assert abs(float(diff) - float(g[f'diff{step}'])) < 1e-6
        assert np.allclose(q.cpu().numpy(), g[f'quant{step}'], atol=1e-6)
        sd = tr.state_dict()
        assert int(sd['quantize.counter']) == int(g[f'counter{step + 1}'])
        assert np.allclose(sd['quantize.ema_cluster_size_hidden'].cpu().numpy(), g[f'cs{step + 1}'], rtol=1e-6, atol=1e-7)
        assert np.allclose(sd['quantize.ema_dw_hidden'].cpu().numpy(), g[f'dw{step + 1}'], rtol=1e-5, atol=1e-6)
        assert np.allclose(sd['quantize.embeddings'].cpu().numpy(), g[f'E{step + 1}'], rtol=2e-5, atol=1e-6)
    tr.training = False                                       # eval: lookup only, the codebook stays put
    E = tr.embeddings.clone()
    tr(torch.from_numpy(g['z0']).to(dev))
    assert torch.equal(E, tr.embeddings)


def test_quantizer_ema_accumulate_is_deterministic_and_additive_over_replicas(dev):
    """full-size codebook (256 x 1024): counts / sums of a batch == the sum over its two halves (what the all-reduce of
    utils_th.py:50-52 relies on), bit-reproducible run to run, and equal to the one-hot matmul of the reference"""
    from viewformer_amd import _lib, ops
    from viewformer_amd.ops import _p, _stream
    lib = _lib.load()
    D, Kc, M = 256, 1024, 14336
    g = np.random.Generator(np.random.PCG64(9))
    z = torch.from_numpy((g.standard_normal((M, D)) * 0.3).astype(np.float32)).to(dev)
    idx = torch.from_numpy(g.integers(0, 300, size=M)).to(dev)          # skewed: many empty codes, some heavy ones

    def acc(zz, ii):
        c = torch.full((Kc,), float('nan'), device=dev)
        s = torch.full((D, Kc), float('nan'), device=dev)
        _lib.check(lib.vf_vq_ema_accumulate_f32(_p(zz), _p(ii), zz.shape[0], D, Kc, _p(c), _p(s), _stream()), 'acc')
        return c, s
    c, s = acc(z, idx)
    c2, s2 = acc(z, idx)
    assert torch.equal(c, c2) and torch.equal(s, s2)
    onehot = torch.nn.functional.one_hot(idx, Kc).double()
    assert torch.equal(c.double(), onehot.sum(0))
    ref = z.double().t() @ onehot
    assert ((s.double() - ref).abs().max() / ref.abs().max()).item() < 1e-6
    ca, sa = acc(z[:M // 2].contiguous(), idx[:M // 2].contiguous())
    cb, sb = acc(z[M // 2:].contiguous(), idx[M // 2:].contiguous())
    assert torch.equal(ca + cb, c)
    assert ((sa + sb - s).abs().max() / s.abs().max()).item() < 1e-6


@pytest.mark.gpu
def test_resize_u8_bit_identical_to_reference_golden_and_oracle():
    """vf_resize_u8 (evaluators' pre-process resize, data/_common.py:19-61) vs the reference-recorded outputs, and vs the oracle on
    more size pairs; empty batch"""
    import os
    from oracle import vqgan_oracle as vq
    from viewformer_amd import ops
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'resize.npz'))
    for i in range(8):
        dst, meth = (int(v) for v in g[f'meta{i}'])
        method = None if meth < 0 else ['nearest', 'bilinear'][meth]
        got = ops.resize_u8(torch.from_numpy(g[f'in{i}']).cuda(), dst, method).cpu().numpy()
        assert np.array_equal(got, g[f'out{i}']), i
    rng = np.random.default_rng(3)
    for src, dst in [(129, 128), (131, 64), (512, 128), (100, 128), (17, 128), (255, 32)]:
        img = rng.integers(0, 256, size=(2, src, src, 3), dtype=np.uint8)
        got = ops.resize_u8(torch.from_numpy(img).cuda(), dst).cpu().numpy()
        assert np.array_equal(got, vq.resize_u8(img, dst)), (src, dst)
    assert ops.resize_u8(torch.zeros((0, 64, 64, 3), dtype=torch.uint8, device='cuda'), 128).shape == (0, 128, 128, 3)


@pytest.mark.gpu
def test_empty_batches_are_not_errors():
    """the reference returns empty tensors for an empty batch (torch semantics); so does the HIP model"""
    from viewformer_amd.config import VQGANConfig
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.weights import make_vqgan_weights
    cfg = VQGANConfig(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], image_size=32, z_channels=32, embed_dim=32, n_embed=64)
    model = VQGAN(cfg, data_format='NCHW', device='cuda').load_state_dict(make_vqgan_weights(cfg, seed=3))
    quant, diff, codes = model.encode(torch.zeros((0, 3, 32, 32), device='cuda'))
    assert codes.shape == (0, 16, 16) and codes.dtype == torch.int64 and quant.shape == (0, 32, 16, 16)
    assert model.decode_code(codes).shape == (0, 3, 32, 32)


@pytest.mark.gpu
@pytest.mark.parametrize('n,chunk', [(600, 150), (1040, 260)])
def test_encoder_large_launches_match_chunked_launches(dev, full_vq, n, chunk):
    """size-independent property at full size: one launch chunk of n images (activation tensors past 2^32 bytes at n = 600 and past
    2^31 elements at n = 1040) gives bit-identical codes to the same images in chunks — no 32-bit index anywhere on the path"""
    from viewformer_amd.weights import synthetic_scene_batch
    cfg, sd, g = full_vq
    frames, _ = synthetic_scene_batch(2, 8, 128, seed=77)
    base = torch.from_numpy(frames.reshape(16, 128, 128, 3))
    imgs = base.repeat((n + 15) // 16, 1, 1, 1)[:n].contiguous().to(dev)
    imgs[1::3] = imgs[1::3].flip(1)                                      # not just 16 distinct images
    imgs[2::5] = imgs[2::5].flip(2)
    from viewformer_amd.vqgan import VQGAN
    big = VQGAN(cfg, data_format='NHWC', max_images_per_call=n).load_state_dict(sd).to(dev)
    small = VQGAN(cfg, data_format='NHWC', max_images_per_call=chunk).load_state_dict(sd).to(dev)
    c_big = big.encode(imgs)[-1]
    c_small = small.encode(imgs)[-1]
    assert c_big.shape == (n, 8, 8) and torch.equal(c_big, c_small)
    dec_big = big.decode_code(c_big[:n // 2])
    dec_small = small.decode_code(c_big[:n // 2])
    assert torch.equal(dec_big, dec_small)


@pytest.mark.gpu
def test_evaluators_resize_frames_of_another_size(dev, tiny_vq):
    """evaluate_codebook.py:67-77 / evaluate_transformer.py:105: frames that are not image_size go through the reference's resize
    (here on the GPU) before the encoder: same codes as encoding the oracle-resized frames"""
    from oracle import vqgan_oracle as vq
    from viewformer_amd.evaluate import codebook_batch_predictions
    cfg, sd, g = tiny_vq
    m = _vq_model(cfg, sd, dev, 'NHWC')
    rng = np.random.default_rng(9)
    big = rng.integers(0, 256, size=(3, 80, 80, 3), dtype=np.uint8)
    out = codebook_batch_predictions(m, torch.from_numpy(big))
    small = vq.resize_u8(big, cfg.image_size)
    ref = codebook_batch_predictions(m, torch.from_numpy(small))
    assert torch.equal(out['codes'], ref['codes']) and torch.equal(out['generated_images'], ref['generated_images'])
    assert out['ground_truth_images'].shape == (3, 80, 80, 3)             # the ground truth is returned as given (:75)


@pytest.mark.parametrize('HW,C,n', [(256, 256, 5), (64, 512, 3), (64, 256, 2)])
def test_fused_spatial_attention_matches_fp64_and_batched_form(dev, HW, C, n):
    """vf_attn_spatial_f32 (AttnBlock core in one kernel, scores on chip) against fp64 softmax(q k^T C^-0.5) v and against the batched
    igemm + row-softmax path it replaces (vqgan_th.py:124-141)"""
    from viewformer_amd import ops
    g = np.random.Generator(np.random.PCG64(HW + C))
    qkv = torch.from_numpy((g.standard_normal((n * HW, 3 * C)) * 0.7).astype(np.float32)).to(dev)
    scale = float(int(C) ** (-0.5))
    got = ops.attn_spatial(qkv, n, HW, C, scale)
    q, k, v = [t.double().cpu().view(n, HW, C) for t in (qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:])]
    ref = torch.softmax(q @ k.transpose(1, 2) * scale, -1) @ v
    err = (got.cpu().double().view(n, HW, C) - ref).abs().max().item()
    assert err < 5e-6 * max(1.0, ref.abs().max().item()), err
    # the unfused product path
    kp = ops.pack(qkv[:, C:2 * C], C, HW, 1, sk=1, sn=3 * C, st=0, batch=n, src_bstride=HW * 3 * C)
    S = torch.empty((n, HW, HW), dtype=torch.float32, device=dev)
    ops.igemm(qkv[:, :C], kp, HW, C, HW, S, lda=3 * C, batch=n, stride_x=HW * 3 * C, stride_w=ops.packed_floats(C, HW), stride_out=HW * HW)
    ops.softmax_rows_(S, n * HW, HW, scale)
    vp = ops.pack(qkv[:, 2 * C:], HW, C, 1, sk=3 * C, sn=1, st=0, batch=n, src_bstride=HW * 3 * C)
    a = torch.empty((n * HW, C), dtype=torch.float32, device=dev)
    ops.igemm(S, vp, HW, HW, C, a, lda=HW, batch=n, stride_x=HW * HW, stride_w=ops.packed_floats(HW, C), stride_out=HW * C)
    assert (got - a).abs().max().item() < 5e-6 * max(1.0, ref.abs().max().item())
    assert not ops.attn_spatial_supported(128, 256)
    # the x3h form (fp16 matrix pipe, three exact products per fp32 product): fp32-equivalent — its error against fp64 is not above the
    # native kernel's (+ the representation floor), on unit-scale inputs and on inputs whose softmax is peaked (tiny probabilities)
    for mul in (1.0, 4.0):
        qk = qkv.clone()
        qk[:, :2 * C] *= mul ** 0.5
        q, k, v = [t.double().cpu().view(n, HW, C) for t in (qk[:, :C], qk[:, C:2 * C], qk[:, 2 * C:])]
        ref2 = torch.softmax(q @ k.transpose(1, 2) * scale, -1) @ v
        e32 = (ops.attn_spatial(qk, n, HW, C, scale).cpu().double().view(n, HW, C) - ref2).abs().max().item()
        e3h = (ops.attn_spatial(qk, n, HW, C, scale, x3h=True).cpu().double().view(n, HW, C) - ref2).abs().max().item()
        print(f'attn_spatial HW={HW} C={C} x{mul}: f32 err {e32:.2e}  x3h err {e3h:.2e}')
        assert e3h < 1.25 * e32 + 2e-7 * max(1.0, ref2.abs().max().item()), (e3h, e32)
