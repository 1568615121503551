"""GPU: the filtered codebook lookup (csrc/vq_filter.hip: fp16 candidate filter + exact fp32 re-rank) returns the indices of the exact
f32-MFMA kernel (csrc/vq_argmin.hip) BIT FOR BIT — on the reference-recorded golden, on millions of random rows at every scale, and
on rows constructed to hit each escape path (ambiguous window, lost candidates -> full scan, out-of-range / non-finite rows)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs the MI355X'
    return torch.device('cuda:0')


def _codebook(seed=0, scale=0.05, D=256, Kc=1024):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(((g.random((D, Kc)) * 2 - 1) * np.sqrt(3.0) * scale).astype(np.float32))


def _both(z, E, dev, stats=True):
    from viewformer_amd import ops
    E = E.to(dev)
    z = z.to(dev).contiguous()
    D, Kc = E.shape
    Ep, esq = ops.vq_pack_codebook(E)
    exact = ops.vq_argmin(z, Ep, esq, D, Kc)
    blob = ops.vq_filter_pack(E)
    st = torch.zeros(4, dtype=torch.int32, device=dev) if stats else None
    filt = ops.vq_argmin_filtered(z, blob, D, Kc, stats=st)
    return exact, filt, (st.cpu().numpy() if stats else None)


def test_filtered_lookup_matches_reference_golden(dev):
    from viewformer_amd.config import VQGANConfig
    from viewformer_amd.weights import make_vqgan_weights
    g = load_golden('vq_lookup.npz')
    sd = make_vqgan_weights(VQGANConfig(), seed=int(g['seed']), codebook_scale=float(g['codebook_scale']))
    E = torch.from_numpy(np.asarray(sd['quantize.embeddings']))
    z = torch.from_numpy(g['z']).permute(0, 2, 3, 1).reshape(-1, 256)
    exact, filt, st = _both(z, E, dev)
    assert torch.equal(exact, filt)
    ref = torch.from_numpy(g['idx']).reshape(-1)
    bad = (filt.cpu() != ref)
    assert bad.sum() == 0 or (g['margin'].reshape(-1)[bad.numpy()] == 0).all()     # the exact-midpoint row: either tied code
    assert st.sum() - st[2] == -(-z.shape[0] // 128) * 128                           # every row took exactly one path


@pytest.mark.parametrize('scale,M', [(1.0, 200000), (0.2, 57344), (0.01, 30001), (30.0, 30000), (1e-5, 4096), (300.0, 4096)])
def test_filtered_equals_exact_on_random_rows(dev, scale, M):
    """size-independent property at bench size and beyond (57 344 rows = one 896-image encoder launch): identical indices"""
    g = np.random.Generator(np.random.PCG64(int(scale * 1000) + M))
    z = torch.from_numpy((g.standard_normal((M, 256)) * scale).astype(np.float32))
    E = _codebook(seed=1)
    exact, filt, st = _both(z, E, dev)
    nbad = int((exact != filt).sum())
    print(json.dumps(dict(test='vq_filter_random', scale=scale, rows=M, certified=int(st[0]), reranked=int(st[1]), exact_evals=int(st[2]),
                          scanned=int(st[3]), mismatches=nbad)))
    assert nbad == 0
    assert int(st[0]) + int(st[1]) + int(st[3]) == -(-M // 128) * 128             # every row (tail rows of the last tile included) took one path


def test_filtered_equals_exact_on_encoder_like_rows_and_reports_rates(dev, full_vq):
    """z as the encoder produces it (the 20k-token golden frames): identical indices; the filter certifies most rows alone"""
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.weights import synthetic_scene_batch
    cfg, sd, _ = full_vq
    frames, _ = synthetic_scene_batch(8, 8, 128, seed=41)
    x = torch.from_numpy(frames.reshape(-1, 128, 128, 3)).to(dev)
    m_f = VQGAN(cfg, data_format='NHWC', lookup='filter').load_state_dict(sd).to(dev)
    m_e = VQGAN(cfg, data_format='NHWC', lookup='exact').load_state_dict(sd).to(dev)
    r_f, r_e = m_f.encode(x), m_e.encode(x)
    assert m_f._E_filter is not None and m_e._E_filter is None
    assert torch.equal(r_f[-1], r_e[-1])
    exact, filt, st = _both(r_f._z, torch.from_numpy(np.asarray(sd['quantize.embeddings'])), dev)
    assert torch.equal(exact, filt) and torch.equal(filt.view_as(r_f[-1]), r_f[-1])
    rep = dict(test='vq_filter_encoder_rows', rows=int(filt.numel()), certified=int(st[0]), reranked=int(st[1]), exact_evals=int(st[2]),
               scanned=int(st[3]))
    print(json.dumps(rep))
    try:
        with open(os.path.join(REPO, 'gpurun_out', 'parity_report.jsonl'), 'a') as f:
            f.write(json.dumps(rep) + '\n')
    except OSError:
        pass
    assert st[0] > 0.6 * filt.numel() and st[3] == 0


def test_filtered_lookup_escape_paths(dev):
    """duplicated codes in ONE lane (>= 3 candidates a lane cannot hold -> exact scan, lowest index wins), exact midpoints, rows that
    are codes, a zero row, rows beyond fp16's range, inf / nan rows, and a ragged row count"""
    E = _codebook(seed=3)
    for k in (37, 69, 101, 997):                       # codes = 5 mod 32 -> the same lane of the filter's accumulator layout
        E[:, k] = E[:, 5]
    E[:, 640] = E[:, 77]                               # a duplicate in another lane pair (two candidates: re-rank path)
    g = np.random.Generator(np.random.PCG64(8))
    M = 1000 + 37
    z = torch.from_numpy((g.standard_normal((M, 256)) * 0.1).astype(np.float32))
    z[0] = E[:, 5]                                     # 5-fold tie -> scan -> index 5
    z[1] = E[:, 77] * 1.0000001                        # 2-fold tie -> re-rank -> index 77
    z[2] = 0.5 * (E[:, 3] + E[:, 900])                 # exact midpoint
    z[3] = 0
    z[4] = E[:, 5] + 1e-4 * torch.from_numpy(g.standard_normal(256).astype(np.float32))
    z[5] = 7.0e4                                       # beyond fp16: scanned exactly
    z[6, 10] = float('inf')
    z[7, 3] = float('nan')
    z[8] = -E[:, 11] * 3
    z[M - 1] = E[:, 1023]
    exact, filt, st = _both(z, E, dev)
    finite = torch.ones(M, dtype=torch.bool)
    finite[6] = finite[7] = False
    assert torch.equal(exact.cpu()[finite], filt.cpu()[finite])
    assert int(filt[0]) == 5 and int(filt[1]) == 77 and int(filt[M - 1]) == 1023 and int(filt[4]) == 5
    assert int(filt[2]) in (3, 900)
    assert 0 <= int(filt[6]) < 1024 and 0 <= int(filt[7]) < 1024        # non-finite rows: defined, in-range output (no crash)
    assert st[3] >= 3                                                   # the out-of-range row and the non-finite rows are scanned
    assert st[2] >= 32                                                  # the 5-fold tie in one lane re-ranks that lane's whole column
    # an empty batch and unsupported shapes
    from viewformer_amd import ops, _lib
    blob = ops.vq_filter_pack(E.to(dev))
    assert ops.vq_argmin_filtered(torch.empty((0, 256), device=dev), blob, 256, 1024).numel() == 0
    assert not ops.vq_filter_supported(128, 1024) and not ops.vq_filter_supported(256, 2048) and not ops.vq_filter_supported(256, 1000)
    with pytest.raises(_lib.VfError):
        ops.vq_filter_pack(torch.zeros((128, 1024), device=dev))


@pytest.mark.parametrize('kind', ['beyond_fp16', 'inf_entry'])
def test_codebook_outside_fp16_range_takes_the_exact_scan(dev, kind):
    """the fp16 tiles cannot carry a code entry >= 65504 (or a non-finite one); the pack step then leaves e_max = inf,
    no row is certified and every row is scanned exactly — the same indices as vq_argmin instead of silently diverging."""
    g = np.random.Generator(np.random.PCG64(5))
    E = _codebook(seed=3)
    if kind == 'beyond_fp16':
        E[7, 100] = 1.0e5
        E[9, 101] = -7.0e4
    else:
        E[3, 55] = float('inf')
    z = torch.from_numpy(g.standard_normal((640, 256)).astype(np.float32))
    exact, filt, st = _both(z, E, dev)
    assert torch.equal(exact, filt)
    assert st[3] == 640 and st[0] == 0 and st[1] == 0      # stats[3] = rows scanned exactly over the whole codebook
