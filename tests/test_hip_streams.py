"""GPU: the multi-stream ("branching") forward — attention STREAMS mask, MIGT with output_poses /
localization_tokens / compute_losses, and the multi-context evaluator — against the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import TINY_MIGT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs the MI355X'
    return torch.device('cuda:0')


def _maxerr(a, b):
    return (a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max().item()


def _rand(shape, seed, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


@pytest.mark.parametrize('B,H,S,L,NS', [(2, 2, 3, 64, 3), (1, 1, 4, 16, 2), (1, 3, 7, 64, 3)])
def test_attention_streams_mask_equals_branching_attention(dev, B, H, S, L, NS):
    """kernel STREAMS mode == compute_causal_block_multiend_attention (branching_attention.py:82-126)"""
    from oracle import migt_oracle as mg
    from viewformer_amd import ops
    d = H * 64
    T = NS * S * L
    qkv = _rand((B * T, 3 * d), 5, 0.35)
    x = qkv.double().view(B, NS, S, L, 3 * d)
    ks, vs, qs = [], [], []
    for s in range(NS):
        v, q, k = x[:, s].chunk(3, -1)
        ks.append(mg._split_heads(k, H)); vs.append(mg._split_heads(v, H)); qs.append(mg._split_heads(q, H))
    ref = torch.stack([mg._merge_heads(a) for a in mg.compute_causal_block_multiend_attention(ks, vs, qs)], 1)
    g = qkv.to(dev)
    outs = []
    for skip in (True, False):
        out = torch.empty((B * T, d), device=dev)
        ops.attn_blockcausal(g[:, d:2 * d], g[:, 2 * d:], g[:, :d], out, B, H, T, L, 3 * d, 3 * d, 3 * d, d, 1.0, skip, -S)
        err = _maxerr(out.view(B, NS, S, L, d), ref)
        assert err < 3e-5, (skip, err)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])


def _tiny(loc, seed=2):
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.weights import make_migt_weights
    cfg = MIGTConfig(**TINY_MIGT, localization_weight='1' if loc else '0', pose_multiplier=0.2)
    sd = make_migt_weights(cfg, seed=seed, std=0.08)
    return cfg, sd


@pytest.mark.parametrize('loc', [False, True])
def test_migt_training_graph_forward_matches_oracle(dev, loc):
    """compute_losses=True builds the 2-/3-stream graph (migt.py:371-401); forward values only"""
    from oracle import migt_oracle as mg
    from viewformer_amd.migt import MIGT
    from viewformer_amd.weights import synthetic_scene_batch
    cfg, sd = _tiny(loc)
    g = np.random.Generator(np.random.PCG64(3))
    B, S, t = 2, 4, cfg.token_image_size
    ids = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(B, S, t, t)))
    _, cams = synthetic_scene_batch(B, S, 8, 4)
    cams = mg.normalize_cameras(torch.from_numpy(cams))
    m = MIGT(cfg).load_state_dict(sd).to(dev)
    out = m(dict(input_ids=ids.to(dev), poses=cams.to(dev)), compute_losses=True)
    ref = mg.migt_forward(sd, cfg, ids, cams, dtype=torch.float64, compute_losses=True)
    assert len(out['hidden_states']) == (3 if loc else 2)
    assert _maxerr(out['logits'], ref['logits']) < 2e-4
    for a, b in zip(out['hidden_states'], ref['hidden_states']):
        assert _maxerr(a, b) < 2e-4
    if loc:
        assert _maxerr(out['pose_prediction'], ref['pose_prediction']) < 2e-4
    with pytest.raises(RuntimeError, match='MIGTTrainer'):                  # the training graph lives in train.MIGTTrainer
        m(dict(input_ids=ids.to(dev), poses=cams.to(dev)), training=True)


def test_multictx_streams_match_oracle_and_single_stream_calls(dev):
    """output_poses + localization_tokens (evaluate_transformer_multictx.py:60-73): the MASK stream at position i
    equals a plain single-stream call on [views < i, MASK], bit-class fp32."""
    from oracle import migt_oracle as mg
    from viewformer_amd.migt import MIGT
    from viewformer_amd.weights import synthetic_scene_batch
    cfg, sd = _tiny(True, seed=6)
    g = np.random.Generator(np.random.PCG64(8))
    B, S, t = 2, 4, cfg.token_image_size
    codes = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(B, S, t, t)))
    _, cams = synthetic_scene_batch(B, S, 8, 5)
    cams = mg.normalize_cameras(torch.from_numpy(cams))
    ids = torch.cat([codes[:, :-1], torch.full_like(codes[:, :1], cfg.n_embeddings)], 1)
    ctx = torch.cat([cams[:, :-1], torch.zeros_like(cams[:, :1])], 1)
    qcam = cams[:, -1:].expand(B, S, 7).contiguous()
    qtok = codes[:, -1:].expand(B, S, t, t).contiguous()
    m = MIGT(cfg).load_state_dict(sd).to(dev)
    out = m(dict(input_ids=ids.to(dev), poses=ctx.to(dev), localization_tokens=qtok.to(dev), output_poses=qcam.to(dev)))
    ref = mg.migt_forward(sd, cfg, ids, ctx, localization_tokens=qtok, output_poses=qcam, dtype=torch.float64)
    assert _maxerr(out['logits'], ref['logits']) < 2e-4
    assert _maxerr(out['pose_prediction'], ref['pose_prediction']) < 2e-4
    # position S-1 of the MASK stream == the ordinary generation call
    single = m(dict(input_ids=ids.to(dev), poses=cams.to(dev)), last_view_logits_only=True)
    assert _maxerr(out['logits'][:, -1], single['logits_last'].cpu()) < 1e-5


def test_multictx_evaluator_end_to_end(dev, tiny_vq):
    """shape/protocol check of the multi-context evaluator + agreement of its last position with the
    single-context evaluator on the same scene"""
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.evaluate import generate_batch_predictions as single_ctx
    from viewformer_amd.evaluate_multictx import generate_batch_predictions as multi_ctx
    from viewformer_amd.migt import MIGT
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    vcfg, vsd, _ = tiny_vq                                  # 32 px images -> 16x16 tokens
    mcfg = MIGTConfig(n_embeddings=vcfg.n_embed, n_head=2, d_model=128, n_layer=2, token_image_size=16,
                      sequence_size=3, localization_weight='1', pose_multiplier=0.2)
    msd = make_migt_weights(mcfg, seed=9, std=0.08)
    vq = VQGAN(vcfg, data_format='NHWC').load_state_dict(vsd).to(dev)
    tr = MIGT(mcfg).load_state_dict(msd).to(dev)
    frames, cams = synthetic_scene_batch(2, 3, 32, seed=4)
    out = multi_ctx(tr, vq, frames, cams)
    assert tuple(out['generated_images'].shape) == (2, 3, 32, 32, 3) and out['generated_images'].dtype == torch.uint8
    assert tuple(out['generated_cameras'].shape) == (2, 3, 7)
    one = single_ctx(tr, vq, frames, cams, return_codes=True)
    same = (out['generated_codes'][:, -1] == one['generated_codes']).float().mean().item()
    assert same > 0.99, same
    assert _maxerr(out['generated_cameras'][:, -1], one['generated_cameras'].cpu()) < 1e-4
