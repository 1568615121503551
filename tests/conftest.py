import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """every GPU test gets a wall-clock limit (pytest-timeout, thread method: the process is ended even when the main thread sits inside a
    HIP call).  A kernel fault surfaces as an abort on some boxes and as a synchronise that never returns on others (round 3 lost 15
    GPU-minutes to one): a bounded failure instead of a hung box."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker('gpu') is not None and item.get_closest_marker('timeout') is None:
            item.add_marker(pytest.mark.timeout(600, method='thread'))


def parity_report(**kw):
    """one line of gpurun_out/parity_report.jsonl (the per-round copy lives under profiles/) and on stdout"""
    import json
    try:
        os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(REPO, 'gpurun_out', 'parity_report.jsonl'), 'a') as f:
            f.write(json.dumps(kw, default=str) + '\n')
    except OSError:
        pass
    print(json.dumps(kw, default=str))


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


TINY_VQ = dict(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], image_size=32,
               z_channels=32, embed_dim=32, n_embed=64)
TINY_MIGT = dict(n_embeddings=64, n_head=2, d_model=128, n_layer=2, token_image_size=4, sequence_size=4)


@pytest.fixture(scope='session')
def tiny_vq():
    from viewformer_amd.config import VQGANConfig
    from viewformer_amd.weights import make_vqgan_weights
    g = load_golden('vqgan_tiny.npz')
    cfg = VQGANConfig(**TINY_VQ)
    sd = make_vqgan_weights(cfg, seed=int(g['seed']), codebook_scale=float(g['codebook_scale']))
    return cfg, sd, g


@pytest.fixture(scope='session')
def full_vq():
    from viewformer_amd.config import VQGANConfig
    from viewformer_amd.weights import make_vqgan_weights
    g = load_golden('vqgan_full.npz')
    cfg = VQGANConfig()
    sd = make_vqgan_weights(cfg, seed=int(g['seed']), codebook_scale=float(g['codebook_scale']))
    return cfg, sd, g
