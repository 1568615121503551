"""GPU: parity at scale (VERDICT r1 'next' #1).

* token flip rate on 20 480 tokens recorded from the reference's own ``VQGAN.encode`` (tests/golden/vqgan_codes_20k.npz,
  make_codes_golden.py), per arithmetic arm, every mismatch listed with the reference's own top-2 margin;
* an end-to-end assertion for the MIXED arm — the one ``bench.py`` times — down to generated codes and uint8 pixels;
* BASELINE configs[2]: 19 context views + localization head (S = 20), tiny width against the oracle, full size against the
  oracle and through size-independent properties.
Each test appends its measured numbers to ``gpurun_out/parity_report.jsonl`` (copied to profiles/ per round)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import REPO, TINY_MIGT, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs the MI355X'
    return torch.device('cuda:0')


def _report(**kw):
    try:
        os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(REPO, 'gpurun_out', 'parity_report.jsonl'), 'a') as f:
            f.write(json.dumps(kw) + '\n')
    except OSError:
        pass
    print(json.dumps(kw))


def _maxerr(a, b):
    return (a.detach().cpu().double() - torch.as_tensor(b).detach().cpu().double()).abs().max().item()


# stated bound for an admissible flip: the build's z differs from the reference's by <= 5e-5 per element (asserted in
# test_hip_models.py); a code pair (a, b) then moves by at most 2 * |dz| . |e_a - e_b| <= 2 * sqrt(256) * 5e-5 * |e_a - e_b|, and
# |e_a - e_b| <= 2 * 0.05 * sqrt(3) * 16 for this codebook: 4.5e-3 worst case.  Measured flips sit 100x below that (margins ~1e-5);
# the test admits a flip only (i) to the reference's own runner-up and (ii) when the reference's fp32 top-2 margin is below
FLIP_MARGIN_MAX = 2e-4
FLIP_RATE_MAX = 1e-3        # and at most 0.1 % of tokens


@pytest.mark.parametrize('arith', ['f32', 'x6', 'x3h', 'x3h/32x32x16'])
def test_token_flip_rate_on_20k_reference_tokens(dev, arith):
    """'x3h' runs the stride-1 convolutions on the 16x16x32 MFMA kernel (the default), 'x3h/32x32x16' on the 32x32x16 one
    (vf_select(VF_SEL_CONV_X3H_K32, 0)): two accumulation orders, the same tokens"""
    from viewformer_amd import _lib
    k32 = 0 if arith.endswith('32x32x16') else 1
    _lib.select(_lib.SEL_CONV_X3H_K32, k32)
    try:
        _token_flip_rate(dev, arith.split('/')[0], arith)
    finally:
        _lib.select(_lib.SEL_CONV_X3H_K32, 1)


def _token_flip_rate(dev, arith, label):
    from viewformer_amd.config import VQGANConfig
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.weights import make_vqgan_weights, synthetic_scene_batch
    g = load_golden('vqgan_codes_20k.npz')
    cfg = VQGANConfig()
    sd = make_vqgan_weights(cfg, seed=int(g['seed']), codebook_scale=float(g['codebook_scale']))
    m = VQGAN(cfg, data_format='NHWC', conv_arith=arith).load_state_dict(sd).to(dev)
    frames, _ = synthetic_scene_batch(int(g['n_scenes']), int(g['n_views']), 128, seed=int(g['input_seed']))
    frames = torch.from_numpy(frames.reshape(-1, 128, 128, 3)).to(dev)
    codes = m.encode(frames)[-1].cpu().numpy()
    ref, runner, margin = g['codes'].astype(np.int64), g['runner_up'].astype(np.int64), g['margin']
    assert codes.shape == ref.shape and ref.size >= 16384
    bad = codes != ref
    flips = [dict(token=int(i), ref=int(ref.reshape(-1)[i]), got=int(codes.reshape(-1)[i]), runner_up=int(runner.reshape(-1)[i]),
                  ref_margin=float(margin.reshape(-1)[i])) for i in np.flatnonzero(bad.reshape(-1))]
    _report(test='token_flip_rate', arith=label, tokens=int(ref.size), flips=len(flips), flip_rate=len(flips) / ref.size,
            tokens_with_ref_margin_below_1e_4=int((margin < 1e-4).sum()), min_ref_margin=float(margin.min()), detail=flips)
    for f in flips:
        assert f['got'] == f['runner_up'], f'flip to a code that is not the reference runner-up: {f}'
        assert f['ref_margin'] < FLIP_MARGIN_MAX, f'flip at a non-degenerate margin: {f}'
    assert len(flips) <= FLIP_RATE_MAX * ref.size, f'{len(flips)} flips in {ref.size} tokens'
    # the chunked launch and one big launch agree bit for bit (batch invariance at this size)
    m.max_images_per_call = 64
    assert np.array_equal(m.encode(frames)[-1].cpu().numpy(), codes)


# ---------------------------------------------------------------------------------------------- the timed (mixed) arm, end to end
MIXED_LOGIT_TOL_REL = 3e-2      # as tests/test_hip_bf16.py
MIXED_U8_TOL_LEVELS = 10        # decoder on bf16 MFMA: final image vs the fp32 oracle's decode of the SAME codes
MIXED_U8_MEAN_LEVELS, MIXED_U8_P99_LEVELS, MIXED_U8_P999_LEVELS = 1.0, 3, 5   # ... and its distribution (measured: mean 0.55, max 6-7)
MIXED_POSE_TOL = 3e-2           # generated camera (position in scene units / unit quaternion)
MIXED_POSE_TOL_WIDE = 1.5e-1    # ... at 1.5x the init scale (std 0.03): measured 1.0e-1


@pytest.mark.parametrize('std', [0.02, 0.03])     # the reference's init scale (flat logits) and 1.5x it (per-layer gain > 1: errors grow
                                                  # with depth; at 3x the init scale a RANDOM 12-layer net is chaotic — 0.43 relative logit
                                                  # error on bf16, measured — which says nothing about a trained one)
def test_mixed_arm_end_to_end_against_oracle(dev, full_vq, std):
    """``bench.py``'s default arm — fp32 (x3h / x6) encoder + lookup, bf16-MFMA transformer and decoder — through
    generate_batch_predictions at the bench's model sizes (12 layers, d = 768, 6 context views + target, localization head)."""
    from oracle import pipeline_oracle as po
    from oracle import vqgan_oracle as vq
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.evaluate import generate_batch_predictions
    from viewformer_amd.migt import MIGT
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    vcfg, vsd, _ = full_vq
    mcfg = MIGTConfig(sequence_size=6, n_loss_skip=1, pose_multiplier=0.2, localization_weight='cosine(0,1,120000)')
    msd = make_migt_weights(mcfg, seed=0, std=std)
    B, S = 2, 7
    frames, cams = synthetic_scene_batch(B, S, 128, seed=21)
    ref = po.generate_batch_predictions(msd, mcfg, vsd, vcfg, frames, cams, return_intermediates=True)
    vq_m = VQGAN(vcfg, data_format='NHWC', decoder_precision='bf16', conv_arith='x3h').load_state_dict(vsd).to(dev)
    tr_m = MIGT(mcfg, precision='bf16', dense_arith='x3h').load_state_dict(msd).to(dev)
    got = generate_batch_predictions(tr_m, vq_m, torch.from_numpy(frames).to(dev), torch.from_numpy(cams).to(dev), return_codes=True)
    assert torch.equal(got['codes'].cpu(), ref['codes'])                         # context + target tokens bit-exact
    rel = ((got['logits_last'].cpu().double() - ref['logits_last'].double()).abs().max() / ref['logits_last'].abs().max()).item()
    same = (got['generated_codes'].cpu() == ref['generated_codes'])
    # where the arg-max differs, the oracle's own logit gap between the two codes must be inside the logit tolerance
    lg = ref['logits_last'].reshape(-1, ref['logits_last'].shape[-1]).double()
    gi, ri = got['generated_codes'].cpu().reshape(-1), ref['generated_codes'].reshape(-1)
    gap = (lg.gather(1, ri.view(-1, 1).long()) - lg.gather(1, gi.view(-1, 1).long())).reshape(-1)
    tol_abs = (MIXED_LOGIT_TOL_REL if std <= 0.02 else 2 * MIXED_LOGIT_TOL_REL) * ref['logits_last'].abs().max().item()
    assert rel < (MIXED_LOGIT_TOL_REL if std <= 0.02 else 2 * MIXED_LOGIT_TOL_REL), rel      # the error grows with the weight scale (measured 9e-3 / 2.6e-2)
    assert (gap <= 2 * tol_abs).all(), f'arg-max differs beyond the logit tolerance: max gap {gap.max().item():.3e} vs {2 * tol_abs:.3e}'
    # decoder: the uint8 image against the fp32 oracle decoding the SAME generated codes
    dec_ref = vq.decode_code(vsd, vcfg, got['generated_codes'].cpu())
    u8_ref = vq.postprocess_u8(dec_ref)                                          # NCHW fp32 -> NHWC uint8
    du = (got['generated_images'].cpu().int() - u8_ref.int()).abs()
    assert got['generated_images'].dtype == torch.uint8 and tuple(du.shape) == (B, 128, 128, 3)
    assert du.max() <= MIXED_U8_TOL_LEVELS, du.max()
    # the bound above is the MAXIMUM over 98 304 uint8 values; the distribution behind it ("report the histogram,
    # not just the max") is stated too: mean below one grey level, 99 % of the values within 3 levels, 99.9 % within 5
    hist = torch.bincount(du.reshape(-1), minlength=MIXED_U8_TOL_LEVELS + 1).tolist()
    cdf = np.cumsum(hist) / float(du.numel())
    p99, p999 = int(np.searchsorted(cdf, 0.99)), int(np.searchsorted(cdf, 0.999))
    assert du.float().mean() < MIXED_U8_MEAN_LEVELS and p99 <= MIXED_U8_P99_LEVELS and p999 <= MIXED_U8_P999_LEVELS, (hist, p99, p999)
    # and against the all-oracle image where the generated codes agree
    full = (got['generated_images'].cpu().int() - ref['generated_images'].int()).abs().float()
    #  ... asserted, not only reported: in every scene whose 64 generated tokens ALL agree with the all-oracle generation, the
    # distance to the ALL-ORACLE image — the oracle's encoder, transformer, arg-max and decoder, nothing of the GPU's in it — has the bounded
    # distribution stated above.  (Scene-level on purpose: the decoder's attention blocks are global, one differing token anywhere moves every
    # pixel — measured on this test: on cells whose own token and 8 neighbours agree but another token of the scene differs, the all-oracle
    # image is a DIFFERENT image, mean 10.7 levels.)  At init-scale weights few scenes agree in all 64 tokens; the count is reported.
    agree_scene = same.reshape(B, -1).all(1)
    p999_full = None
    if bool(agree_scene.any()):
        fv = full[agree_scene].reshape(-1)
        cdf_full = np.cumsum(torch.bincount(fv.long(), minlength=MIXED_U8_TOL_LEVELS + 1).tolist()) / float(fv.numel())
        p999_full = int(np.searchsorted(cdf_full, 0.999))
        assert fv.max() <= MIXED_U8_TOL_LEVELS and p999_full <= MIXED_U8_P999_LEVELS and fv.mean() < MIXED_U8_MEAN_LEVELS, (p999_full, float(fv.mean()))
    e_cam = _maxerr(got['generated_cameras'], ref['generated_cameras'])
    # the camera head's error grows with the weight scale like the logits' (measured 0.10 at std 0.03, round 4: reported then, asserted now
    # with its own stated bound — 5x the init-scale bound, one unit-quaternion component / scene unit in seven)
    assert e_cam < (MIXED_POSE_TOL if std <= 0.02 else MIXED_POSE_TOL_WIDE), e_cam
    _report(test='mixed_arm_end_to_end', weight_std=std, scenes=B, views=S, logit_rel_err=rel,
            max_abs_logit=float(ref['logits_last'].abs().max()), generated_code_agreement=float(same.float().mean()),
            max_logit_gap_at_disagreement=float(gap.max()), u8_max_diff_same_codes=int(du.max()),
            u8_mean_diff_same_codes=float(du.float().mean()), u8_diff_histogram_same_codes=hist, u8_p99=p99, u8_p999=p999,
            u8_mean_diff_vs_full_oracle=float(full.mean()), u8_p999_vs_full_oracle_in_fully_agreeing_scenes=p999_full,
            fully_agreeing_scenes=int(agree_scene.sum()), camera_err=e_cam)
    if std <= 0.02:
        assert same.float().mean() > 0.9, same.float().mean()


# ---------------------------------------------------------------------------------------------- the mixed arm on PEAKED logits
PEAKED_MIN_MAX_LOGIT = 10.0     # the trained head must be at least this peaked for the check to mean anything
PEAKED_CODE_AGREEMENT = 0.99    # bf16 arm vs the fp32-equivalent arm on the trained model, generated tokens
PEAKED_LOGIT_TOL_REL = 3e-2     # as MIXED_LOGIT_TOL_REL


def test_mixed_arm_on_a_trained_model_with_peaked_logits(dev, full_vq):
    """every mixed-arm tolerance so far was checked on the flat logits of a random-init head (max |logit| 2-3.6).
    No trained checkpoint can be fetched offline, so the test makes one: the full-size MIGT is trained with the repo's own MIGTTrainer
    (bf16 arm, viewformer/models/migt.py:464-505) on a handful of synthetic scenes until its masked-view head is confident, and the
    timed (bf16) inference arm is then compared with the fp32-equivalent arm — and, for one scene, with the fp64 oracle — on those
    decision margins.  Every disagreeing token is reported with the fp32 arm's own top-2 gap."""
    from oracle import migt_oracle as mg
    from viewformer_amd import geometry
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    vcfg, vsd, _ = full_vq
    B, S = 8, 7
    frames, cams = synthetic_scene_batch(B, S, 128, seed=33)
    vq_m = VQGAN(vcfg, data_format='NHWC', conv_arith='x3h').load_state_dict(vsd).to(dev)
    codes = vq_m.encode(torch.from_numpy(frames.reshape(-1, 128, 128, 3)).to(dev))[-1].view(B, S, 8, 8)
    del vq_m
    poses = geometry.normalize_cameras(geometry.to_relative_cameras(torch.from_numpy(cams))[0])
    cfg = MIGTConfig(sequence_size=S, n_loss_skip=1, pose_multiplier=0.2, localization_weight='1', dropout=0.0, learning_rate=3e-4,
                     weight_decay=0.01, total_steps=2000)
    tr = MIGTTrainer(MIGT(cfg, precision='bf16').load_state_dict(make_migt_weights(cfg, seed=0)).to(dev), warmup_steps=20)
    ce, steps = float('inf'), 0
    while ce > 0.3 and steps < 700:
        for _ in range(50):
            met = tr.train_step(poses, codes, reduce_gradients=False)
        steps += 50
        ce = float(met['ce_loss'])
    sd = tr.state_dict()
    del tr
    torch.cuda.empty_cache()
    ids = torch.cat([codes[:, :-1], torch.full_like(codes[:, :1], cfg.n_embeddings)], 1).to(torch.int32)
    outs = {}
    for arm in ('f32', 'bf16'):
        m = MIGT(cfg, precision=arm).load_state_dict(sd).to(dev)
        outs[arm] = m(dict(input_ids=ids.to(dev), poses=poses.to(dev)), last_view_logits_only=True)['logits_last'].float().cpu()
        del m
    l32, l16 = outs['f32'].reshape(-1, 1024).double(), outs['bf16'].reshape(-1, 1024).double()
    peak = float(l32.abs().max())
    top2 = torch.topk(l32, 2, dim=1)
    margin = (top2.values[:, 0] - top2.values[:, 1])
    c32, c16 = l32.argmax(1), l16.argmax(1)
    bad = (c32 != c16).nonzero().reshape(-1)
    rel = float((l16 - l32).abs().max() / l32.abs().max())
    detail = [dict(token=int(i), f32=int(c32[i]), bf16=int(c16[i]), f32_top2_margin=float(margin[i]),
                   f32_gap_to_bf16_choice=float(l32[i, c32[i]] - l32[i, c16[i]])) for i in bad]
    # the fp64 oracle on one scene pins the fp32-equivalent arm on the trained weights
    sd_np = {k: np.asarray(v) for k, v in sd.items()}
    ref = mg.migt_forward(sd_np, cfg, ids[:1].long().cpu(), poses[:1].cpu(), dtype=torch.float64)['logits'][:, -1].reshape(-1, 1024)
    e_oracle = float((l32[:64] - ref).abs().max())
    agree_oracle = float((ref.argmax(1) == c32[:64]).float().mean())
    hit = float((c32.view(B, -1) == codes[:, -1].reshape(B, -1).cpu()).float().mean())        # how well the model learnt the target view
    _report(test='mixed_arm_peaked_logits', train_steps=steps, final_ce=ce, max_abs_logit=peak, median_top2_margin=float(margin.median()),
            target_token_accuracy=hit, logit_rel_err_bf16_vs_f32=rel, generated_code_agreement=1.0 - bad.numel() / c32.numel(),
            disagreements=detail, f32_arm_vs_fp64_oracle_logit_err=e_oracle, f32_arm_vs_fp64_oracle_code_agreement=agree_oracle)
    assert peak >= PEAKED_MIN_MAX_LOGIT, f'training did not produce a peaked head: max |logit| {peak:.2f} after {steps} steps (ce {ce:.3f})'
    assert e_oracle < 1e-3 * max(1.0, peak), e_oracle
    ref_margin = torch.topk(ref, 2, dim=1).values
    ref_margin = ref_margin[:, 0] - ref_margin[:, 1]
    assert bool(((ref.argmax(1) == c32[:64]) | (ref_margin < 2 * e_oracle)).all())           # fp32-equivalent arm == oracle outside its own error
    assert rel < PEAKED_LOGIT_TOL_REL, rel
    assert 1.0 - bad.numel() / c32.numel() >= PEAKED_CODE_AGREEMENT, detail
    for d in detail:                                  # a disagreement may only happen inside the logit tolerance
        assert d['f32_gap_to_bf16_choice'] <= 2 * PEAKED_LOGIT_TOL_REL * peak, d


# ---------------------------------------------------------------------------------------------- BASELINE configs[2]: S = 20
def test_s20_tiny_width_matches_oracle_with_localization(dev):
    """19-view context + target (T = 20 views), image + localization heads, tiny width: every output against the fp64 oracle,
    generation pass, localization pass and the fused twin pass"""
    from oracle import migt_oracle as mg
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    kw = dict(TINY_MIGT)
    kw['sequence_size'] = 20
    cfg = MIGTConfig(**kw, localization_weight='5', pose_multiplier=0.2, n_loss_skip=4)       # README.md:231-243 (InteriorNet)
    sd = make_migt_weights(cfg, seed=5, std=0.08)
    B, S, t = 2, 20, cfg.token_image_size
    g = np.random.Generator(np.random.PCG64(23))
    ids = torch.from_numpy(g.integers(0, cfg.n_embeddings, size=(B, S, t, t)))
    _, cams = synthetic_scene_batch(B, S, 8, 8)
    cams = mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(cams))[0])
    m = MIGT(cfg).load_state_dict(sd).to(dev)
    gen_ids = torch.cat([ids[:, :-1], torch.full_like(ids[:, :1], cfg.n_embeddings)], 1)
    out = m(dict(input_ids=gen_ids.to(dev), poses=cams.to(dev)))
    ref = mg.migt_forward(sd, cfg, gen_ids, cams, dtype=torch.float64)
    assert _maxerr(out['logits'], ref['logits']) < 3e-4
    out2 = m(dict(input_ids=ids.to(dev), poses=cams[:, :-1].to(dev)))
    ref2 = mg.migt_forward(sd, cfg, ids, cams[:, :-1], dtype=torch.float64)
    assert _maxerr(out2['pose_prediction'], ref2['pose_prediction']) < 3e-4
    lg, pose = m.generate_and_localize(ids.to(dev), cams.to(dev))
    # (bit-identity of the fused pass is a property of L = 64 tiles — asserted at full size below; with L = 16 four views share a
    # 64-row tile and the twin views change its summation order)
    assert _maxerr(lg, out['logits'][:, -1]) < 1e-5 and _maxerr(pose, out2['pose_prediction'][:, -1:]) < 1e-5


def test_s20_full_size_matches_oracle_and_properties(dev):
    """full-size MIGT (12 layers, d = 768, L = 64) at S = 20 (T = 1280 tokens; 1344 in the fused twin pass): one scene against the
    fp64 oracle, then size-independent properties on a batch: fused == two passes, dense == tile-skipping, and a 20-view scene's first
    7 views are bit-identical to the same scene truncated to 7 views (block-causality)"""
    from oracle import migt_oracle as mg
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    cfg = MIGTConfig(sequence_size=20, n_loss_skip=4, localization_weight='5', pose_multiplier=0.2)
    sd = make_migt_weights(cfg, seed=0)
    B, S = 3, 20
    g = np.random.Generator(np.random.PCG64(29))
    ids = torch.from_numpy(g.integers(0, 1024, size=(B, S, 8, 8)))
    _, cams = synthetic_scene_batch(B, S, 8, 12)
    cams = mg.normalize_cameras(mg.to_relative_cameras(torch.from_numpy(cams))[0])
    for arm, tol in (('f32', 1e-3), ('bf16', None)):
        m = MIGT(cfg, precision=arm).load_state_dict(sd).to(dev)
        gen_ids = torch.cat([ids[:, :-1], torch.full_like(ids[:, :1], cfg.n_embeddings)], 1)
        o1 = m(dict(input_ids=gen_ids.to(dev), poses=cams.to(dev)), last_view_logits_only=True)
        o2 = m(dict(input_ids=ids.to(dev), poses=cams[:, :-1].to(dev)), last_view_logits_only=True)
        lg, pose = m.generate_and_localize(ids.to(dev), cams.to(dev))
        assert torch.equal(lg, o1['logits_last']) and torch.equal(pose, o2['pose_prediction'])        # fused twin pass, T = 21 views
        r1 = mg.migt_forward(sd, cfg, gen_ids[:1], cams[:1], dtype=torch.float64)
        r2 = mg.migt_forward(sd, cfg, ids[:1], cams[:1, :-1], dtype=torch.float64)
        e_l = _maxerr(lg[:1], r1['logits'][:, -1])
        e_p = _maxerr(pose[:1], r2['pose_prediction'][:, -1:])
        scale = r1['logits'][:, -1].abs().max().item()
        _report(test='s20_full_size', arm=arm, logit_err=e_l, logit_max=scale, pose_err=e_p)
        if tol is not None:
            assert e_l < tol and e_p < tol
        else:
            assert e_l < 3e-2 * scale and e_p < 3e-2
        # dense "-1e4" form == masked-tile skipping: a property of each kernel that has both forms (attention_f32 / attention_lp).  The
        # bf16 arm's product kernel (attention_dma.hip) only has the skipping form and, since round 4, its own rounding points: compare
        # the two forms on the register-staged kernel, and the product kernel against it within the arm's tolerance
        from viewformer_amd import _lib
        m_dense = MIGT(cfg, precision=arm, skip_masked=False).load_state_dict(sd).to(dev)
        lg_dense = m_dense(dict(input_ids=gen_ids.to(dev), poses=cams.to(dev)), last_view_logits_only=True)['logits_last']
        prev = _lib.select(_lib.SEL_ATTN_DMA, 0)
        try:
            lg_skip = m(dict(input_ids=gen_ids.to(dev), poses=cams.to(dev)), last_view_logits_only=True)['logits_last']
        finally:
            _lib.select(_lib.SEL_ATTN_DMA, prev)
        assert torch.equal(lg_dense, lg_skip)
        if arm == 'bf16':
            assert _maxerr(o1['logits_last'], lg_skip) < 1e-2 * scale
        else:
            assert torch.equal(o1['logits_last'], lg_skip)
        # block-causality: views 0..6 of the 20-view pass == the 7-view pass
        full = m(dict(input_ids=ids.to(dev), poses=cams.to(dev)))['hidden_states'][0]
        short = m(dict(input_ids=ids[:, :7].to(dev), poses=cams[:, :7].to(dev)))['hidden_states'][0]
        assert torch.equal(full[:, :7], short)
