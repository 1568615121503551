#!/usr/bin/env python
"""CPU experiment (VERDICT r2 'next' #1a): would a Winograd F(2x2, 3x3) form of the encoder's stride-1 3x3 convolutions keep the
token indices of the 20 480-token reference golden?

The stride-1, padding-1 3x3 convolutions of the ENCODER (vqgan_th.py:60-70,78-90; conv_in with its 3 input channels stays direct) are
replaced inside the oracle's encoder by an emulation of the kernel a gfx950 build would run:

  V = B^T d B      input transform of each 4x4 patch (stride 2), fp32, one rounding per add
  U = G g G^T      weight transform in fp64, then either rounded to fp32 ('wino_f32') or scaled by a power of two and split into
                   fp16 pieces h + 2^-11 l' exactly as viewformer_amd's x3h packing does ('wino_x3h')
  M[e] = V[e] U[e] 16 element GEMMs over the input channels; 'wino_x3h': three fp32-accumulated products of fp16-valued operands
                   (each product exact in fp32), cross terms in their own accumulator, combined once; 'wino_f32': one fp32 matmul
  Y = A^T M A      output transform, fp32, one rounding per add; then bias (and the caller's residual add)

and the resulting z goes through the reference's distance expression and arg-min.  Reported per arm: flips against the codes the
reference itself recorded (tests/golden/vqgan_codes_20k.npz), max / rms |z - z_direct|, and — what a margin-certified escalation
needs — the distribution of the change of every token's top-2 gap (d_runner_up - d_best of the REFERENCE's pair, evaluated in fp64 on
each arm's z) relative to the direct fp32 oracle.

Run in the build container:  python tools/wino_flip_probe.py [--images 320] [--min-res 0] > profiles/r3_wino_flip_probe.txt
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import vqgan_oracle as vo                                                   # noqa: E402
from viewformer_amd.config import VQGANConfig                                           # noqa: E402
from viewformer_amd.weights import make_vqgan_weights, synthetic_scene_batch            # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)


def split_f16(x64, scale_to=None):
    """x = h + 2^-11 l' with h = rne_f16(x), l' = rne_f16((x - h) 2^11): the x3h operand split."""
    x = x64.to(torch.float32)
    h = x.to(torch.float16)
    l = ((x - h.to(torch.float32)) * 2048.0).to(torch.float16)
    return h.to(torch.float32), l.to(torch.float32)


def in_transform(d):
    """d [..., 4, 4] fp32 -> B^T d B, fp32 adds in a fixed order (rows, then columns)."""
    r0, r1, r2, r3 = d[..., 0, :], d[..., 1, :], d[..., 2, :], d[..., 3, :]
    t = torch.stack([r0 - r2, r1 + r2, r2 - r1, r1 - r3], dim=-2)
    c0, c1, c2, c3 = t[..., 0], t[..., 1], t[..., 2], t[..., 3]
    return torch.stack([c0 - c2, c1 + c2, c2 - c1, c1 - c3], dim=-1)


def out_transform(m):
    """m [..., 4, 4] fp32 -> A^T m A [..., 2, 2], fp32 adds in a fixed order."""
    r0, r1, r2, r3 = m[..., 0, :], m[..., 1, :], m[..., 2, :], m[..., 3, :]
    t = torch.stack([(r0 + r1) + r2, (r1 - r2) - r3], dim=-2)
    c0, c1, c2, c3 = t[..., 0], t[..., 1], t[..., 2], t[..., 3]
    return torch.stack([(c0 + c1) + c2, (c1 - c2) - c3], dim=-1)


class Wino:
    def __init__(self, arm, min_res):
        self.arm, self.min_res, self.cache, self.layers = arm, min_res, {}, set()

    def weights(self, sd, name):
        if name not in self.cache:
            g = vo._t(sd, name + '.weight', torch.float64)                         # [Cout, Cin, 3, 3]
            u = torch.einsum('ea,oiab,fb->efio', G, g, G).reshape(16, g.shape[1], g.shape[0])   # [16, Cin, Cout] fp64
            if self.arm == 'wino_x3h':
                s = 2.0 ** np.floor(np.log2(2.0 ** 14 / u.abs().max().item() * (1 - 1e-12)))      # max|u| s in [2^13, 2^14)
                h, l = split_f16(u * s)
                self.cache[name] = (h, l, 1.0 / s)
            else:
                self.cache[name] = (u.to(torch.float32),)
        return self.cache[name]

    def conv(self, sd, name, x):
        n, c, hh, ww = x.shape
        self.layers.add((name, c, hh))
        d = F.unfold(F.pad(x, (1, 1, 1, 1)), kernel_size=4, stride=2)              # [n, c*16, tiles]
        tiles = d.shape[-1]
        d = d.reshape(n, c, 4, 4, tiles).permute(0, 4, 1, 2, 3)                    # [n, tiles, c, 4, 4]
        v = in_transform(d).reshape(n * tiles, c, 16).permute(2, 0, 1).contiguous()  # [16, n*tiles, c]
        w = self.weights(sd, name)
        if self.arm == 'wino_x3h':
            wh, wl, inv_s = w
            vh, vl = split_f16(v)
            main = torch.bmm(vh, wh)
            cross = torch.bmm(vl, wh) + torch.bmm(vh, wl)                           # second accumulator (fp32)
            m = (main + cross * (2.0 ** -11)) * inv_s
        else:
            m = torch.bmm(v, w[0])
        cout = m.shape[-1]
        m = m.permute(1, 2, 0).reshape(n, tiles, cout, 4, 4)
        y = out_transform(m)                                                        # [n, tiles, cout, 2, 2]
        th, tw = hh // 2, ww // 2
        y = y.reshape(n, th, tw, cout, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(n, cout, hh, ww)
        return y + vo._t(sd, name + '.bias', torch.float32).view(1, -1, 1, 1)


def run(arm, sd, cfg, frames, min_res, batch=8):
    orig = vo.conv
    wino = Wino(arm, min_res) if arm != 'direct' else None

    def conv(sd_, name, x, dtype, stride=1, padding=0):
        w = sd_[name + '.weight']
        if (wino is not None and name.startswith('encoder.') and stride == 1 and padding == 1 and tuple(w.shape[2:]) == (3, 3)
                and w.shape[1] % 32 == 0 and x.shape[-1] >= min_res and x.shape[-1] % 2 == 0):
            return wino.conv(sd_, name, x)
        return orig(sd_, name, x, dtype, stride=stride, padding=padding)
    vo.conv = conv
    zs = []
    try:
        with torch.no_grad():
            for i in range(0, frames.shape[0], batch):
                x = vo.preprocess_u8(torch.from_numpy(frames[i:i + batch]))
                zs.append(vo.encode_z(sd, cfg, x))
    finally:
        vo.conv = orig
    return torch.cat(zs), (sorted(wino.layers) if wino else [])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=320)
    ap.add_argument('--min-res', type=int, default=0, help='only convolutions on maps at least this wide take the Winograd form')
    ap.add_argument('--arms', default='direct,wino_f32,wino_x3h')
    a = ap.parse_args()
    torch.set_num_threads(8)
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'vqgan_codes_20k.npz'))
    cfg = VQGANConfig()
    sd = make_vqgan_weights(cfg, seed=int(g['seed']), codebook_scale=float(g['codebook_scale']))
    sd = {k: torch.as_tensor(v) for k, v in sd.items()}
    frames, _ = synthetic_scene_batch(int(g['n_scenes']), int(g['n_views']), 128, seed=int(g['input_seed']))
    frames = frames.reshape(-1, 128, 128, 3)[:a.images]
    ref = torch.from_numpy(g['codes'].astype(np.int64))[:a.images].reshape(-1)
    runner = torch.from_numpy(g['runner_up'].astype(np.int64))[:a.images].reshape(-1)
    margin = g['margin'][:a.images].reshape(-1)
    emb64 = sd['quantize.embeddings'].double()                                      # [D, K]

    def gap64(z):
        """fp64 distance gap d(runner_up) - d(best) of the REFERENCE's pair on this arm's z."""
        f = z.permute(0, 2, 3, 1).reshape(-1, z.shape[1]).double()
        ea, eb = emb64[:, ref].T, emb64[:, runner].T
        return ((f - eb) ** 2).sum(1) - ((f - ea) ** 2).sum(1)

    out = {}
    z_direct = gap_direct = None
    for arm in a.arms.split(','):
        t0 = time.time()
        z, layers = run(arm, sd, cfg, frames, a.min_res)
        codes = vo.quantize(sd, z)[2].reshape(-1)
        bad = (codes != ref).nonzero().reshape(-1)
        rec = dict(arm=arm, images=int(frames.shape[0]), tokens=int(ref.numel()), seconds=round(time.time() - t0, 1),
                   flips=int(bad.numel()),
                   flip_detail=[dict(token=int(i), ref=int(ref[i]), got=int(codes[i]), runner_up=int(runner[i]),
                                     ref_margin=float(margin[i])) for i in bad],
                   winograd_layers=[f'{n} C={c} {r}x{r}' for n, c, r in layers])
        gp = gap64(z)
        if arm == 'direct':
            z_direct, gap_direct = z, gp
        else:
            dz = (z - z_direct).double()
            dg = (gp - gap_direct).abs()
            rec.update(z_max_abs_diff_vs_direct=float(dz.abs().max()), z_rms_diff_vs_direct=float(dz.pow(2).mean().sqrt()),
                       gap_change_max=float(dg.max()), gap_change_p999=float(dg.quantile(0.999)), gap_change_p99=float(dg.quantile(0.99)),
                       gap_change_rms=float(dg.pow(2).mean().sqrt()),
                       tokens_with_ref_margin_below={f'{t:g}': int((margin < t).sum()) for t in (1e-5, 2e-5, 3e-5, 5e-5, 1e-4, 2e-4)},
                       images_with_a_token_below={f'{t:g}': int((margin.reshape(-1, 64) < t).any(1).sum()) for t in (1e-5, 2e-5, 3e-5, 5e-5, 1e-4, 2e-4)})
        out[arm] = rec
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
