#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference, which never travels to
the GPU box):   python tests/golden/make_golden.py

It imports the reference's PyTorch VQGAN (viewformer/models/vqgan_th.py +
utils_th.py) with three stub modules for packages the image lacks
(aparse.Literal, pytorch_lightning.LightningModule, lpips.LPIPS — none of them
is on the forward path), loads the build's deterministic synthetic weights into
it with ``load_state_dict`` and records inputs -> outputs of the reference's own
``encode`` / ``decode_code`` / ``QuantizeEMA.forward``.  Only data is written:
inputs, expected outputs, and the seeds/configs that regenerate the weights.

The MIGT transformer is TensorFlow-only and cannot be imported here (parity
unpinned, see oracle/migt_oracle.py), so there is no golden for it.
"""
import os
import sys
import types
import typing

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from viewformer_amd.config import VQGANConfig                      # noqa: E402
from viewformer_amd.weights import make_vqgan_weights, synthetic_scene_batch   # noqa: E402

TINY = dict(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16], image_size=32,
            z_channels=32, embed_dim=32, n_embed=64)
CODEBOOK_SCALE = 0.05


def import_reference():
    ap = types.ModuleType('aparse')
    ap.Literal = typing.Literal
    sys.modules['aparse'] = ap
    pl = types.ModuleType('pytorch_lightning')
    pl.LightningModule = torch.nn.Module
    pl.LightningDataModule = object
    sys.modules['pytorch_lightning'] = pl
    lp = types.ModuleType('lpips')

    class LPIPS(torch.nn.Module):
        def __init__(self, net=None):
            super().__init__()
    lp.LPIPS = LPIPS
    sys.modules['lpips'] = lp
    sys.path.insert(0, '/root/reference')
    from viewformer.models import AutoModelTH
    from viewformer.models.config import VQGANConfig as RefCfg
    return AutoModelTH, RefCfg


def build_reference(AutoModelTH, RefCfg, cfg_kwargs, seed):
    ref = AutoModelTH.from_config(RefCfg(**cfg_kwargs)).eval()
    cfg = VQGANConfig(**cfg_kwargs)
    sd = make_vqgan_weights(cfg, seed=seed, codebook_scale=CODEBOOK_SCALE)
    ref.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return ref, cfg, sd


def margins(ref, z):
    """top-2 margin of -dist for every row, computed with the reference's own expression."""
    q = ref.quantize
    x = z.permute(0, 2, 3, 1)
    f = x.reshape(-1, x.size(-1))
    dist = f.pow(2).sum(1, keepdim=True) - 2 * f @ q.embeddings + q.embeddings.pow(2).sum(0, keepdim=True)
    t = torch.topk(-dist, 2, dim=1).values
    return (t[:, 0] - t[:, 1]).numpy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    AutoModelTH, RefCfg = import_reference()
    out = {}

    # ---- tiny config: everything recorded --------------------------------------------
    ref, cfg, _ = build_reference(AutoModelTH, RefCfg, TINY, seed=3)
    frames, _ = synthetic_scene_batch(1, 6, TINY['image_size'], seed=5)
    x = (torch.from_numpy(frames[0]).float() * torch.tensor(1.0 / 255) * 2 - 1).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        z = ref.quant_conv(ref.encoder(x))
        quant, diff, codes = ref.encode(x)
        dec = ref.decode_code(codes)
        fwd = ref(x)[0]
    out['tiny'] = dict(frames=frames[0], z=z.numpy(), codes=codes.numpy(), quant=quant.numpy(),
                       diff=np.float32(diff.item()), decoded=dec.numpy(), forward=fwd.numpy(),
                       margin=margins(ref, z))
    np.savez_compressed(os.path.join(HERE, 'vqgan_tiny.npz'), seed=3, input_seed=5, codebook_scale=CODEBOOK_SCALE,
                        **out['tiny'])

    # ---- full 128px config (BASELINE config #1: batch 4, 128x128) ----------------------
    ref, cfg, _ = build_reference(AutoModelTH, RefCfg, {}, seed=0)
    frames, _ = synthetic_scene_batch(1, 4, 128, seed=0)
    x = (torch.from_numpy(frames[0]).float() * torch.tensor(1.0 / 255) * 2 - 1).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        z = ref.quant_conv(ref.encoder(x))
        quant, diff, codes = ref.encode(x)
        dec = ref.decode_code(codes)
    np.savez_compressed(os.path.join(HERE, 'vqgan_full.npz'), seed=0, input_seed=0, codebook_scale=CODEBOOK_SCALE,
                        z=z.numpy(), codes=codes.numpy(), diff=np.float32(diff.item()),
                        decoded=dec[:2].numpy().astype(np.float32), margin=margins(ref, z),
                        decoded_mean=dec.mean(dim=(1, 2, 3)).numpy(), decoded_absmax=dec.abs().amax(dim=(1, 2, 3)).numpy())

    # ---- codebook lookup alone (QuantizeEMA.forward, eval) ---------------------------------
    g = np.random.Generator(np.random.PCG64(77))
    zz = torch.from_numpy(g.standard_normal((8, 256, 8, 8)).astype(np.float32) * np.float32(0.2))
    # adversarial rows: exact duplicates of codes, midpoints of two codes, and exact ties
    E = ref.quantize.embeddings
    zt = zz.permute(0, 2, 3, 1).reshape(-1, 256).clone()
    zt[0] = E[:, 17]
    zt[1] = 0.5 * (E[:, 3] + E[:, 900])
    zt[2] = 0
    zz = zt.reshape(8, 8, 8, 256).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        q, d, idx = ref.quantize(zz)
    np.savez_compressed(os.path.join(HERE, 'vq_lookup.npz'), seed=0, codebook_scale=CODEBOOK_SCALE,
                        z=zz.numpy(), idx=idx.numpy(), quant=q.numpy(), diff=np.float32(d.item()),
                        margin=margins(ref, zz))
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
