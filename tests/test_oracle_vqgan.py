"""CPU: pin oracle/vqgan_oracle.py against the golden vectors recorded from the
reference's own VQGAN (tests/golden/make_golden.py)."""
import numpy as np
import torch

from conftest import load_golden
from oracle import vqgan_oracle as vq
from viewformer_amd.weights import synthetic_scene_batch, make_vqgan_weights
from viewformer_amd.config import VQGANConfig


def _x(frames):
    return vq.preprocess_u8(torch.from_numpy(frames))


def test_tiny_encode_matches_reference(tiny_vq):
    cfg, sd, g = tiny_vq
    frames, _ = synthetic_scene_batch(1, 6, cfg.image_size, seed=int(g['input_seed']))
    assert np.array_equal(frames[0], g['frames'])          # the input generator is portable
    x = _x(frames[0])
    z = vq.encode_z(sd, cfg, x)
    assert np.allclose(z.numpy(), g['z'], atol=1e-5, rtol=1e-5)   # utils/testing.py:98 tolerance
    quant, diff, codes = vq.encode(sd, cfg, x)
    assert codes.dtype == torch.int64
    assert np.array_equal(codes.numpy(), g['codes'])       # bit-exact indices
    assert np.allclose(quant.numpy(), g['quant'], atol=1e-6)
    assert abs(float(diff) - float(g['diff'])) < 1e-6


def test_tiny_decode_matches_reference(tiny_vq):
    cfg, sd, g = tiny_vq
    dec = vq.decode_code(sd, cfg, torch.from_numpy(g['codes']))
    assert np.allclose(dec.numpy(), g['decoded'], atol=1e-5, rtol=1e-5)
    # VQGAN.forward == decode(quant) with the straight-through value == decode_code(codes)
    assert np.allclose(dec.numpy(), g['forward'], atol=1e-5, rtol=1e-5)


def test_full_encode_decode_matches_reference(full_vq):
    cfg, sd, g = full_vq
    frames, _ = synthetic_scene_batch(1, 4, 128, seed=int(g['input_seed']))
    x = _x(frames[0])
    z = vq.encode_z(sd, cfg, x)
    assert np.allclose(z.numpy(), g['z'], atol=2e-5, rtol=1e-5)
    codes = vq.quantize(sd, z)[-1]
    assert np.array_equal(codes.numpy(), g['codes'])
    assert len(np.unique(g['codes'])) > 20                 # the lookup is not degenerate
    dec = vq.decode_code(sd, cfg, torch.from_numpy(g['codes'][:2]))
    assert np.allclose(dec.numpy(), g['decoded'], atol=2e-5, rtol=1e-5)


def test_lookup_matches_reference_including_ties():
    g = load_golden('vq_lookup.npz')
    sd = make_vqgan_weights(VQGANConfig(), seed=int(g['seed']), codebook_scale=float(g['codebook_scale']))
    q, diff, idx = vq.quantize(sd, torch.from_numpy(g['z']))
    assert np.array_equal(idx.numpy(), g['idx'])
    assert idx.reshape(-1)[0] == 17                          # a row equal to code 17 maps to 17
    assert np.allclose(q.numpy(), g['quant'])
    assert abs(float(diff) - float(g['diff'])) < 1e-6


def test_fp64_arm_agrees_where_margin_is_clear(full_vq):
    """The fp64 arm may only disagree with the reference on near-ties."""
    cfg, sd, g = full_vq
    z = torch.from_numpy(g['z'])
    idx64 = vq.quantize(sd, z, dtype=torch.float64)[-1].numpy().reshape(-1)
    ref = g['codes'].reshape(-1)
    bad = idx64 != ref
    assert (g['margin'][bad] < 1e-4).all()


def test_pre_post_process_semantics():
    u8 = torch.arange(256, dtype=torch.uint8).reshape(1, 16, 16, 1).repeat(1, 1, 1, 3)
    x = vq.preprocess_u8(u8)
    assert x.shape == (1, 3, 16, 16) and x.min() == -1 and x.max() == 1
    back = vq.postprocess_u8(x)
    assert torch.equal(back, u8)                             # 255.5 scale + truncation round-trips every level
    y = vq.postprocess_u8(torch.tensor([-3.0, -1.0, 0.0, 0.999, 1.0, 7.0]).reshape(1, 1, 1, 6).repeat(1, 3, 1, 1))
    assert y[0, 0, :, 0].tolist() == [0, 0, 127, 255, 255, 255]


def test_quantizer_ema_training_branch_matches_reference_golden():
    """QuantizeEMA.forward with self.training (utils_th.py:46-64): three consecutive steps recorded from the reference"""
    import os
    import numpy as np
    import torch
    from oracle import vqgan_oracle as vq
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'vq_ema.npz'))
    D, K = g['E0'].shape
    state = dict(embeddings=torch.from_numpy(g['E0']), ema_cluster_size_hidden=torch.zeros(K), ema_dw_hidden=torch.zeros(D, K), counter=0)
    for step in range(3):
        q, diff, ind = vq.quantize_train_step(state, torch.from_numpy(g[f'z{step}']), float(g['decay']), float(g['eps']))
        assert np.array_equal(ind.numpy(), g[f'ind{step}'])
        assert abs(float(diff) - float(g[f'diff{step}'])) < 1e-6
        assert np.allclose(q.numpy(), g[f'quant{step}'], atol=1e-6)
        assert state['counter'] == int(g[f'counter{step + 1}'])
        assert np.allclose(state['ema_cluster_size_hidden'].numpy(), g[f'cs{step + 1}'], rtol=1e-6, atol=1e-7)
        assert np.allclose(state['ema_dw_hidden'].numpy(), g[f'dw{step + 1}'], rtol=1e-5, atol=1e-6)
        assert np.allclose(state['embeddings'].numpy(), g[f'E{step + 1}'], rtol=2e-5, atol=1e-6)


def _summary(t):
    f = np.asarray(t, dtype=np.float64).reshape(-1)
    n = f.size
    return np.array([np.linalg.norm(f), f.sum(), f[0], f[n // 2], f[n - 1]])


def test_vqgan_training_step_oracle_matches_reference_golden():
    """loss terms, the gradient of every parameter, the EMA codebook after the forward and two Adam steps vs the reference"""
    import os
    from oracle import vqgan_oracle as vq
    from oracle import vqgan_train_oracle as vt
    from viewformer_amd.config import VQGANConfig
    from viewformer_amd.weights import make_vqgan_weights
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'vqgan_train_tiny.npz'))
    cfg = VQGANConfig(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], image_size=32, z_channels=32, embed_dim=32,
                      n_embed=64, perceptual_weight=0.0, codebook_weight=1.0, learning_rate=1e-3)
    sd = make_vqgan_weights(cfg, seed=int(g['seed']), codebook_scale=float(g['codebook_scale']))
    x = vq.preprocess_u8(torch.from_numpy(g['frames']))
    names = [str(n) for n in g['param_names']]
    params = {k: np.asarray(sd[k], dtype=np.float64) for k in names}
    bufs = {k: np.asarray(sd[k]) for k in vt.BUFFERS}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v2 = {k: np.zeros_like(v) for k, v in params.items()}
    state = dict(embeddings=torch.from_numpy(np.asarray(bufs['quantize.embeddings'], dtype=np.float32)),
                 ema_cluster_size_hidden=torch.zeros(64), ema_dw_hidden=torch.zeros(32, 64), counter=0)
    for step in (1, 2):
        cur = dict(params)
        cur.update({k: np.asarray(val) for k, val in bufs.items()})
        cur['quantize.embeddings'] = state['embeddings'].numpy()
        grads, metrics, extra = vt.gradients(cur, cfg, x)
        if step == 1:
            assert abs(metrics['loss'] - float(g['loss'])) < 2e-6
            assert abs(metrics['rec_loss'] - float(g['rec_loss'])) < 2e-6 and abs(metrics['quant_loss'] - float(g['quant_loss'])) < 1e-6
            for i, n in enumerate(names):
                got, want = _summary(grads[n].numpy()), g['grad_summary'][i]
                assert np.allclose(got, want, rtol=2e-3, atol=2e-7), (n, got, want)
            for key in g.files:
                if key.startswith('grad:'):
                    assert np.allclose(grads[key[5:]].numpy(), g[key], rtol=2e-3, atol=2e-7), key
        # the forward in training mode moves the codebook (utils_th.py:46-64) ...
        vq.quantize_train_step(state, extra['z'].detach().float(), 0.99, 1e-5)
        if step == 1:
            assert np.allclose(state['embeddings'].numpy(), g['E_after_fwd'], rtol=1e-4, atol=1e-6)
            assert np.allclose(state['ema_cluster_size_hidden'].numpy(), g['cs_after_fwd'], rtol=1e-5, atol=1e-7)
        # ... then Adam moves everything else
        vt.adam_step(params, {k: grads[k].numpy() for k in names}, m, v2, step, float(g['lr']))
        want = g[f'param_summary_step{step}']
        for i, n in enumerate(names):
            if g['grad_summary'][i][0] < 1e-6:
                # a bias in front of a per-channel GroupNorm (32 channels, 32 groups) has a mathematically zero gradient: Adam turns
                # the rounding noise into +-lr steps whose signs no two implementations share — bound the drift instead
                lim = 1.01 * float(g['lr']) * step
                assert np.all(np.abs(params[n] - np.asarray(sd[n], dtype=np.float64)) <= lim), (step, n)
                continue
            # (the first Adam steps are sign-like: elements whose gradient is rounding noise move by +-lr with implementation-
            # dependent sign, so the SUM is not comparable; the norm and the sampled elements are)
            got = _summary(params[n])
            assert np.allclose(got[[0, 2, 3, 4]], want[i][[0, 2, 3, 4]], rtol=1e-4, atol=2.1 * float(g['lr']) * step), (step, n)
            assert abs(got[0] - want[i][0]) < 1e-4 * want[i][0] + 1e-6, (step, n)


def test_resize_oracle_matches_reference_golden():
    """data/_common.py:19-61 resize (nearest up / bilinear down, truncating uint8 cast): bit-identical to the reference's outputs"""
    import os
    from oracle import vqgan_oracle as vq
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'resize.npz'))
    for i in range(8):
        dst, meth = (int(v) for v in g[f'meta{i}'])
        got = vq.resize_u8(g[f'in{i}'], dst, None if meth < 0 else ['nearest', 'bilinear'][meth])
        assert got.dtype == np.uint8 and np.array_equal(got, g[f'out{i}']), i


def test_oracle_matches_reference_codes_on_a_slice_of_the_20k_golden():
    """the CPU restatement against the reference's own recorded tokens beyond the 4-image golden: the first 16 images (1024
    tokens) of tests/golden/vqgan_codes_20k.npz (make_codes_golden.py) — the GPU test covers all 20 480"""
    from conftest import load_golden
    from viewformer_amd.config import VQGANConfig
    from viewformer_amd.weights import make_vqgan_weights, synthetic_scene_batch
    g = load_golden('vqgan_codes_20k.npz')
    cfg = VQGANConfig()
    sd = make_vqgan_weights(cfg, seed=int(g['seed']), codebook_scale=float(g['codebook_scale']))
    frames, _ = synthetic_scene_batch(int(g['n_scenes']), int(g['n_views']), 128, seed=int(g['input_seed']))
    x = vq.preprocess_u8(torch.from_numpy(frames.reshape(-1, 128, 128, 3)[:16]))
    codes = vq.encode(sd, cfg, x)[-1].numpy()
    ref = g['codes'][:16].astype(np.int64)
    bad = codes != ref
    assert bad.sum() == 0 or (g['margin'][:16][bad] < 1e-4).all(), (int(bad.sum()), g['margin'][:16][bad])
    assert (g['runner_up'] != g['codes']).all() and g['codes'].size == 20480 and (g['margin'] >= 0).all()
