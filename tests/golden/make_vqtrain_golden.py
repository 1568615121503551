#!/usr/bin/env python
"""Golden vectors for ONE codebook (VQGAN) training step FROM THE REFERENCE ITSELF (vqgan_th.py: forward :349-352 in training mode,
_compute_loss :354-368 with perceptual_weight = 0, autograd, torch.optim.Adam(lr, betas=(0.5, 0.9)) :427-429).
Run in the build container only:   python tests/golden/make_vqtrain_golden.py
Tiny config + the build's deterministic weights (seed 3, as vqgan_tiny.npz).  Recorded: the loss terms, the gradient of EVERY parameter
reduced to (L2 norm, sum, value at 3 fixed positions) plus a few complete small gradients, the EMA buffers after the forward, and every
parameter's (norm, sum) after two Adam steps.  Only data is written."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import CODEBOOK_SCALE, TINY, build_reference, import_reference      # noqa: E402
from viewformer_amd.weights import synthetic_scene_batch                               # noqa: E402


def summary(t):
    f = t.detach().double().reshape(-1)
    n = f.numel()
    return np.array([f.norm().item(), f.sum().item(), f[0].item(), f[n // 2].item(), f[n - 1].item()], dtype=np.float64)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    AutoModelTH, RefCfg = import_reference()
    kw = dict(TINY, perceptual_weight=0.0, codebook_weight=1.0, learning_rate=1e-3)
    ref, cfg, _ = build_reference(AutoModelTH, RefCfg, kw, seed=3)
    ref.train()
    frames, _ = synthetic_scene_batch(1, 6, TINY['image_size'], seed=5)
    x = (torch.from_numpy(frames[0]).float() * torch.tensor(1.0 / 255) * 2 - 1).permute(0, 3, 1, 2).contiguous()
    opt = ref.configure_optimizers()
    out = dict(frames=frames[0], seed=3, input_seed=5, codebook_scale=CODEBOOK_SCALE, lr=np.float64(1e-3))
    names = [n for n, p in ref.named_parameters() if p.requires_grad]
    out['param_names'] = np.array(names)
    for step in range(2):
        opt.zero_grad()
        xrec, qloss, *_ = ref(x)
        loss, log = ref._compute_loss(qloss, x, xrec, split='train')
        loss.backward()
        if step == 0:
            out['loss'] = np.float64(loss.item())
            out['rec_loss'] = np.float64(log['train/rec_loss'].item())
            out['quant_loss'] = np.float64(log['train/quant_loss'].item())
            out['grad_summary'] = np.stack([summary(dict(ref.named_parameters())[n].grad) for n in names])
            for n in ('encoder.conv_in.bias', 'decoder.conv_out.weight', 'encoder.mid.attn_1.norm.weight', 'quant_conv.bias',
                      'decoder.up.1.upsample.conv.bias', 'encoder.down.0.downsample.conv.bias'):
                out['grad:' + n] = dict(ref.named_parameters())[n].grad.numpy().copy()
            out['E_after_fwd'] = ref.quantize.embeddings.numpy().copy()
            out['cs_after_fwd'] = ref.quantize.ema_cluster_size_hidden.numpy().copy()
        opt.step()
        out[f'param_summary_step{step + 1}'] = np.stack([summary(dict(ref.named_parameters())[n]) for n in names])
        out[f'loss_step{step + 1}'] = np.float64(loss.item())
    np.savez_compressed(os.path.join(HERE, 'vqgan_train_tiny.npz'), **out)
    print(len(names), 'parameters; loss', out['loss'], 'rec', out['rec_loss'], 'quant', out['quant_loss'])


if __name__ == '__main__':
    main()
