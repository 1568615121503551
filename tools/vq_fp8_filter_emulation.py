#!/usr/bin/env python
"""CPU emulation: would an e4m3 (fp8) candidate filter keep the codebook lookup's re-rank set small?  (VERDICT r4 "next" #8.)

The lookup (csrc/vq_filter.hip; reference QuantizeEMA.forward, viewformer/models/utils_th.py:32-44) is a 16-bit MFMA filter with a PROVEN
error window + an exact fp32 re-rank of the codes inside the window: indices bit-exact by construction.  The only way past the fp16
filter's matrix time (12 us at the 2.5 PF peak for 57 344 rows; >= 70 % of the HBM roof would be 10.7 us) is a cheaper filter: the
MX-scaled fp8 MFMAs (K = 64 / 128, block scales E8M0 per 32 elements) run at twice the 16-bit rate on gfx950.  The question this script
answers with data instead of argument: how many codes per row survive an e4m3 filter whose window is still CERTIFIED (so that the result
stays bit-exact), against the fp16 filter's 1.2?  "Build it if the re-rank set stays <= 2x today's."

Emulated filters (scores s~ = sum_d q(z_d) q(E_dk) - ee_k / 2 accumulated in fp64 — the accumulation error is not the issue):
  fp16      today's filter (operands rounded to fp16), window 2 eps from the bound in vq_filter.hip's header
  e4m3-t    per-ROW power-of-two scale on z and per-TENSOR scale on E, both cast to OCP e4m3fn (non-MX MFMA: 1x rate; shown for reference)
  e4m3-mx   per-32-element-block power-of-two (E8M0) scales on both operands along d — the operand format of the 2x-rate MX MFMAs
            (block scales only help elements that would otherwise fall below e4m3's normal range, 2^-14 of the row maximum: on these
            data that is almost none, so the two fp8 filters come out alike — the limit is the 3-bit mantissa, not the exponent range)
Windows for the fp8 filters, from loosest-certified to not certifiable at all:
  cs        Cauchy-Schwarz on the operands' relative rounding error u = 2^-4: eps = (2u + u^2) |z| |e|_max  (the fp16 filter's form)
  norms     the MEASURED rounding residuals: eps_k = |dz| |q(e_k)| + |z| |de_k| with dz = z - q(z) known when z is loaded and |de_k| known
            at pack time (tighter than any a-priori bound; still a proof)
  oracle    NOT a bound: the largest |s~ - s| actually observed for that ROW (unknowable without the exact scores) — the floor no
            certified window can go below
For each: candidates per row (mean / p50 / p99 / max) and the share of rows certified by the filter alone (one candidate).

Data: (a) the 512 reference-recorded rows of tests/golden/vq_lookup.npz (z written by the reference's own encoder); (b) encoder outputs of the
oracle encoder (oracle/vqgan_oracle.py, pinned to the reference by tests/golden) on synthetic frames — the "encoder-like" rows the bench
produces; (c) 200 000 random rows (the scale-1.0 case of tests/test_hip_vq_filter.py, codebook seed 1).

  python tools/vq_fp8_filter_emulation.py [--images 32] > profiles/r5_lookup_fp8_filter_emulation.txt
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def q_fp16(x):
    return x.to(torch.float16).to(torch.float64)


def q_e4m3_scaled(x, amax):
    """cast x * 2^k to e4m3fn with the largest k that keeps amax * 2^k <= 448 (amax broadcastable), back to real values"""
    k = torch.floor(torch.log2(448.0 / amax.clamp_min(1e-30)))
    s = torch.exp2(k)
    y = (x * s).to(torch.float32).to(torch.float8_e4m3fn).to(torch.float64)
    return y / s


def q_e4m3_rowwise(z):
    return q_e4m3_scaled(z, z.abs().amax(1, keepdim=True))


def q_e4m3_tensor(E):
    return q_e4m3_scaled(E, E.abs().max().reshape(1, 1))


def q_e4m3_mx(x, axis):
    """E8M0 block scales over blocks of 32 along ``axis`` (the d axis)"""
    x = x.movedim(axis, -1)
    sh = x.shape
    b = x.reshape(*sh[:-1], sh[-1] // 32, 32)
    y = q_e4m3_scaled(b, b.abs().amax(-1, keepdim=True))
    return y.reshape(sh).movedim(-1, axis)


def fp16_eps(z, E):
    """the window half-width of vq_filter.hip (header + kernel: eps_row), per row"""
    zn = torch.sqrt((q_fp16(z) ** 2).sum(1)) * 1.0005 + 5.0e-7
    e_norm = torch.sqrt((E ** 2).sum(0))
    e_max, ee_max = e_norm.max() * 1.00001, (E ** 2).sum(0).max()
    eps = (zn * e_max) * 1.1300e-3 + ee_max * 7.7e-5 + zn * zn * 6.0e-8 + (zn + e_max) * 5.0e-7 + 1e-12
    return eps * 1.01


def stats(cnt):
    c = cnt.double()
    return dict(mean=round(float(c.mean()), 2), p50=int(c.median()), p99=int(torch.quantile(c, 0.99)), max=int(c.max()),
                certified_alone=round(float((cnt == 1).double().mean()), 4))


def run(name, z, E, chunk=8192):
    z, E = z.double(), E.double()
    D, K = E.shape
    hee = (E ** 2).sum(0) / 2
    e_norm = torch.sqrt((E ** 2).sum(0))
    Ef16, Et, Emx = q_fp16(E), q_e4m3_tensor(E), q_e4m3_mx(E, 0)
    dEt, dEmx = torch.sqrt(((E - Et) ** 2).sum(0)), torch.sqrt(((E - Emx) ** 2).sum(0))
    nEt, nEmx = torch.sqrt((Et ** 2).sum(0)), torch.sqrt((Emx ** 2).sum(0))
    u = 2.0 ** -4
    out = {k: [] for k in ('fp16', 'e4m3-t/cs', 'e4m3-t/norms', 'e4m3-t/oracle', 'e4m3-mx/cs', 'e4m3-mx/norms', 'e4m3-mx/oracle')}
    relerr = {'fp16': 0.0, 'e4m3-t': 0.0, 'e4m3-mx': 0.0}
    wrong = {'e4m3-t': 0, 'e4m3-mx': 0, 'fp16': 0}
    for a in range(0, z.shape[0], chunk):
        zc = z[a:a + chunk]
        s = zc @ E - hee                                        # exact scores (fp64)
        best = s.argmax(1)
        zn = torch.sqrt((zc ** 2).sum(1, keepdim=True))
        N = (zn * e_norm.max()).clamp_min(1e-30)                  # (an all-zero row: every filter is exact there)

        def count(st, eps):                                     # codes inside [max - 2 eps, max]; eps [rows,1] or [rows,K]
            m = st.max(1, keepdim=True).values
            if eps.shape[1] == 1:
                inside = st >= m - 2 * eps
            else:                                               # per-code bound: k survives unless s~_k + eps_k < max_j (s~_j - eps_j)
                lo = (st - eps).max(1, keepdim=True).values
                inside = st + eps >= lo
            return inside.sum(1)

        st = q_fp16(zc) @ Ef16 - hee
        out['fp16'].append(count(st, fp16_eps(zc, E).reshape(-1, 1)))
        relerr['fp16'] = max(relerr['fp16'], float(((st - s).abs().amax(1, keepdim=True) / N).max()))
        wrong['fp16'] += int((st.argmax(1) != best).sum())
        for tag, zq, Eq, dE, nE in (('e4m3-t', q_e4m3_rowwise(zc), Et, dEt, nEt), ('e4m3-mx', q_e4m3_mx(zc, 1), Emx, dEmx, nEmx)):
            st = zq @ Eq - hee
            err = (st - s).abs()
            relerr[tag] = max(relerr[tag], float((err.amax(1, keepdim=True) / N).max()))
            wrong[tag] += int((st.argmax(1) != best).sum())
            out[tag + '/cs'].append(count(st, (2 * u + u * u) * N))
            dz = torch.sqrt(((zc - zq) ** 2).sum(1, keepdim=True))
            out[tag + '/norms'].append(count(st, dz * nE + zn * dE))
            out[tag + '/oracle'].append(count(st, err.amax(1, keepdim=True)))
    res = {k: stats(torch.cat(v)) for k, v in out.items()}
    print(f'## {name}: {z.shape[0]} rows x {K} codes, |z| mean {float(torch.sqrt((z ** 2).sum(1)).mean()):.3f}, |e| max {float(e_norm.max()):.3f}')
    print(f'   largest |s~ - s| / (|z| |e|_max):  fp16 {relerr["fp16"]:.2e}   e4m3-t {relerr["e4m3-t"]:.2e}   e4m3-mx {relerr["e4m3-mx"]:.2e}'
          f'      rows whose FILTER arg-max is not the exact arg-min: fp16 {wrong["fp16"]}, e4m3-t {wrong["e4m3-t"]}, e4m3-mx {wrong["e4m3-mx"]}')
    print(f'   {"filter / window":18s} {"candidates per row: mean":>26s} {"p50":>6s} {"p99":>6s} {"max":>6s}   certified by the filter alone')
    for k, v in res.items():
        print(f'   {k:18s} {v["mean"]:26.2f} {v["p50"]:6d} {v["p99"]:6d} {v["max"]:6d}   {v["certified_alone"]:.4f}')
    return {name: res}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=32, help='synthetic frames encoded by the oracle encoder for data set (b)')
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    from viewformer_amd.config import VQGANConfig
    from viewformer_amd.weights import make_vqgan_weights, synthetic_scene_batch
    from oracle import vqgan_oracle as vq
    allres = {}
    print(__doc__.split('\n\n')[0])
    print()
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'vq_lookup.npz'))
    cfg = VQGANConfig()
    sd = make_vqgan_weights(cfg, seed=int(g['seed']), codebook_scale=float(g['codebook_scale']))
    E = torch.from_numpy(np.asarray(sd['quantize.embeddings']))
    z = torch.from_numpy(g['z']).permute(0, 2, 3, 1).reshape(-1, 256)
    allres.update(run('(a) reference-recorded rows (tests/golden/vq_lookup.npz)', z, E))
    t0 = time.time()
    gf = np.load(os.path.join(REPO, 'tests', 'golden', 'vqgan_full.npz'))
    sdf = make_vqgan_weights(cfg, seed=int(gf['seed']), codebook_scale=float(gf['codebook_scale']))
    frames, _ = synthetic_scene_batch(-(-args.images // 8), 8, 128, seed=41)
    x = vq.preprocess_u8(torch.from_numpy(frames.reshape(-1, 128, 128, 3)[:args.images]))
    zs = []
    with torch.no_grad():
        for a in range(0, x.shape[0], 8):
            zs.append(vq.encode_z(sdf, cfg, x[a:a + 8]))
    zz = torch.cat(zs).permute(0, 2, 3, 1).reshape(-1, 256)
    Ef = torch.from_numpy(np.asarray(sdf['quantize.embeddings']))
    allres.update(run(f'(b) oracle-encoder outputs of {args.images} synthetic 128x128 frames ({time.time() - t0:.0f} s of CPU encode)', zz, Ef))
    rg = np.random.Generator(np.random.PCG64(int(1.0 * 1000) + 200000))
    zr = torch.from_numpy(rg.standard_normal((200000, 256)).astype(np.float32))
    cg = np.random.Generator(np.random.PCG64(1))
    Er = torch.from_numpy(((cg.random((256, 1024)) * 2 - 1) * np.sqrt(3.0) * 0.05).astype(np.float32))
    allres.update(run('(c) 200 000 random rows, scale 1.0 (tests/test_hip_vq_filter.py), codebook seed 1', zr, Er))
    print()
    print('## verdict')
    b = allres[[k for k in allres if k.startswith('(b)')][0]]
    print(f"   encoder-like rows: the fp16 filter leaves {b['fp16']['mean']} candidates per row ({b['fp16']['certified_alone']:.1%} of rows need no re-rank);")
    print(f"   the MX e4m3 filter with the tightest PROVABLE window (measured residual norms) leaves {b['e4m3-mx/norms']['mean']} "
          f"({b['e4m3-mx/norms']['mean'] / max(b['fp16']['mean'], 1e-9):.0f}x), and even the unknowable per-row oracle window leaves "
          f"{b['e4m3-mx/oracle']['mean']}.")
    print('   The bar was <= 2x today\'s re-rank set.  An exact fp32 re-rank costs 256 fmaf per (row, code) on the vector unit: at ~100+ candidates per row it')
    print('   is more arithmetic than the whole fp16 filter executes on the matrix pipe.  e4m3 carries 3 mantissa bits; the distances between the best and')
    print('   the next codes of a 1024-code book in 256 dimensions are ~1e-2 of |z||e|, the fp8 rounding noise ~3e-2 of it.  Not built; DESIGN 6.4 is final.')
    print()
    print(json.dumps(allres))


if __name__ == '__main__':
    main()
