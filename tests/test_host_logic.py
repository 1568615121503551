"""CPU: host-side logic that mirrors the reference's model-object contract — schedule strings (utils/schedules.py), config.json
loading (models/__init__.py:62-78), the dynamic-pose-loss variable in checkpoints (migt.py:105-112,279-280), the evaluators'
resize rule (data/_common.py:19-61)."""
import json
import math

import numpy as np
import pytest

from viewformer_amd import checkpoint as ck
from viewformer_amd.config import MIGTConfig, VQGANConfig, load_config
from viewformer_amd.schedules import parse, Constant, Cosine, Linear, Warmup


def test_schedule_grammar_and_values():
    # constants (ConstantSchedule._from_str, schedules.py:135-142)
    assert parse('1')(10) == 1.0 and parse('5.')(123) == 5.0 and isinstance(parse('0'), Constant)
    # cosine(a,b,N): final + (initial - final) * 0.5 * (cos(min(1, t/N) pi) + 1)   (:201-203)
    c = parse('cosine(0,1,120000)')                                     # README.md:356 (SM7 training command)
    assert isinstance(c, Cosine) and c(0) == 0.0 and abs(c(60000) - 0.5) < 1e-12 and c(120000) == 1.0 and c(10 ** 9) == 1.0
    # linear(a,b,N): initial + min(t/N, 1) * (final - initial)   (:170-171)
    l = parse('linear(2,4,100)')
    assert isinstance(l, Linear) and l(0) == 2.0 and l(25) == 2.5 and l(1000) == 4.0
    # two-argument forms are completed by the model's total_steps (with_total_steps :114-118, migt.py:268)
    l2 = parse('linear(0,2)')
    with pytest.raises(ValueError):
        l2(5)
    assert l2.with_total_steps(1000)(50) == 0.1
    assert parse('cosine(1,0)').with_total_steps(10)(10) == 0.0
    assert parse('cosine(1,0,7)').with_total_steps(10).num_total_steps == 7          # an explicit N is kept
    # warmup(inner, W): (min(t, W) / W) * inner(max(t - W, 0))   (:222-225); parsed from the LAST comma (:240-247)
    w = parse('warmup(cosine(1,0,100),10)')
    assert isinstance(w, Warmup) and w(5) == 0.5 * 1.0 and abs(w(60) - (0.5 * (math.cos(0.5 * math.pi) + 1))) < 1e-12
    assert str(parse('cosine(0,1,120000)')) == 'cosine(0.0,1.0,120000)'              # what asdict() writes back into config.json
    assert str(parse(str(w))) == str(w)
    with pytest.raises(ValueError):
        parse('exponential(1,2)')


def test_is_zero_follows_the_reference():
    # Schedule.is_zero: constant 0, linear/cosine with initial == final == 0, warmup of a zero schedule (:148,183,212,236)
    for s, z in [('0', True), ('0.0', True), ('1', False), ('cosine(0,0,100)', True), ('cosine(0,0)', True), ('linear(0,0,5)', True),
                 ('cosine(0,1,100)', False), ('linear(1,0)', False), ('warmup(cosine(0,0),5)', True), ('warmup(1,5)', False)]:
        assert parse(s).is_zero() is z, s
        assert MIGTConfig(localization_weight=s).use_localization is (not z), s      # migt.py:268-269


def test_load_config_ignores_unknown_keys_like_the_reference():
    cfg = load_config(dict(model='migt', n_layer=3, localization_weight='cosine(0,1)', some_future_field=7, total_steps=50))
    assert isinstance(cfg, MIGTConfig) and cfg.n_layer == 3 and cfg.use_localization
    cfg = load_config(dict(model='vqgan', ch=64, not_a_field='x'))
    assert isinstance(cfg, VQGANConfig) and cfg.ch == 64
    with pytest.raises(ValueError):
        load_config(dict(model='nope'))
    assert load_config(dict(model='migt', localization_weight=0)).use_localization is False   # a JSON number is accepted too


def test_trainer_schedule_parser_accepts_every_reference_form():
    from viewformer_amd.train import parse_schedule
    assert parse_schedule('linear(0,5)', 100)(50) == 2.5
    assert parse_schedule('cosine(0,1)', 120000)(60000) == pytest.approx(0.5)
    assert parse_schedule('warmup(2,10)', 5)(5) == 1.0


@pytest.mark.parametrize('loc', ['1', '0'])
def test_dynamic_pose_loss_checkpoint_round_trip(tmp_path, loc):
    """A Keras checkpoint of a model built with use_dynamic_pose_loss carries pose_loss_weighting_criterion/pos_ori_weights
    (DynamicLossWeightingCriterion, migt.py:105-112, created at :279-280 whether or not the localization head is on); it must load."""
    from dataclasses import asdict
    from viewformer_amd.weights import make_migt_weights
    cfg = MIGTConfig(n_layer=1, d_model=64, n_head=1, localization_weight=loc, use_dynamic_pose_loss=True)
    sd = make_migt_weights(cfg, seed=2)
    assert 'pose_loss_weighting_criterion.pos_ori_weights' in sd
    d = tmp_path / 'tr'
    d.mkdir()
    cj = asdict(cfg)
    cj['model'] = 'migt'
    json.dump(cj, open(d / 'config.json', 'w'))
    keras = ck.state_dict_to_keras(sd)
    assert 'pose_loss_weighting_criterion/pos_ori_weights/.ATTRIBUTES/VARIABLE_VALUE' in keras
    ck.write_tensor_bundle(str(d / 'model'), keras)
    m = ck.load_model(str(d / 'model'))
    got = np.asarray(m.state_dict()['pose_loss_weighting_criterion.pos_ori_weights'])
    assert np.array_equal(got, np.array([0.0, -3.0], np.float32))                    # constant_initializer([0., -3.]), :112
    # without the flag the training-only pair is dropped: the reference's load_model calls load_weights(...).expect_partial()
    # (utils/tensorflow.py:60-62), which tolerates checkpoint values the model has no variable for
    cj['use_dynamic_pose_loss'] = False
    json.dump(cj, open(d / 'config.json', 'w'))
    m2 = ck.load_model(str(d / 'model'))
    assert 'pose_loss_weighting_criterion.pos_ori_weights' not in m2.state_dict()
    # a state dict written before the key was tracked (flag on, key absent) gets the initial value instead of failing strict
    from viewformer_amd.migt import MIGT
    old = {k: v for k, v in sd.items() if not k.startswith('pose_loss_weighting_criterion')}
    m3 = MIGT(cfg).load_state_dict(old)
    assert np.array_equal(np.asarray(m3.state_dict()['pose_loss_weighting_criterion.pos_ori_weights']), np.array([0.0, -3.0], np.float32))
    with pytest.raises(RuntimeError):                                               # any other missing key still raises
        MIGT(cfg).load_state_dict({k: v for k, v in old.items() if k != 'ln_f.gamma'})


def test_resize_rule_is_identity_when_either_side_matches():
    """resize() compares the NHWC batch's shape[-2] (W) and resize_th() the NCHW tensor's shape[-2] (H) with image_size; either
    match returns the frames untouched (data/_common.py:26-27,54-55).  Host-side rule only (no kernel launch on identity)."""
    import torch
    from viewformer_amd import ops
    for shape in [(2, 32, 32, 3), (2, 32, 48, 3), (2, 48, 32, 3)]:
        x = torch.zeros(shape, dtype=torch.uint8)
        assert ops.resize_u8(x, 32) is x
