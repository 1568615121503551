"""ORACLE (test infrastructure, NOT product code) — CPU restatement of one codebook (VQGAN) training step.

Only ``tests/`` may import this module.  It composes the pinned forward restatement (vqgan_oracle.py) into the training graph of
the reference (vqgan_th.py: forward :349-352 in training mode, QuantizeEMA.forward :32-68 incl. the EMA update, ``_compute_loss``
:354-368 with perceptual_weight = 0, torch.optim.Adam(lr, betas=(0.5, 0.9)) :427-429) and takes gradients with torch autograd.
Pinned by tests/golden/vqgan_train_tiny.npz, recorded from the reference itself (tests/golden/make_vqtrain_golden.py).
"""
import numpy as np
import torch

from . import vqgan_oracle as vq

BUFFERS = ('quantize.embeddings', 'quantize.ema_cluster_size_hidden', 'quantize.ema_dw_hidden', 'quantize.counter')


def losses(sd, cfg, x, dtype=torch.float64, lpips_sd=None):
    """-> (loss, dict(rec_loss, quant_loss, z, ind)); x NCHW in [-1, 1]; the codebook used is sd['quantize.embeddings'] (the lookup
    happens BEFORE the EMA update, utils_th.py:36-44).  ``lpips_sd``: frozen LPIPS-VGG weights, required when perceptual_weight > 0
    (vqgan_th.py:402-404; restated in lpips_oracle.py — that part of the graph is parity-unpinned)"""
    x = x.to(dtype)
    z = vq.encode_z(sd, cfg, x, dtype)                                    # encoder + quant_conv, vqgan_th.py:380-381
    with torch.no_grad():
        ind = (-vq.quantize_distances(sd, z, dtype)).max(1)[1].view(z.shape[0], z.shape[2], z.shape[3])
    qe = vq.embed_code(sd, ind, dtype).detach()
    diff = (qe - z).pow(2).mean()                                          # utils_th.py:66  (quantize.detach() - input)
    quant = z + (qe - z).detach()                                          # :67 straight-through
    dec = vq.decoder(sd, cfg, vq.conv(sd, 'post_quant_conv', quant, dtype), dtype)      # vqgan_th.py:385-388 (vq.decode is no_grad)
    rec = (x - dec).abs()                                                  # :401
    p_loss = torch.zeros(1, dtype=dtype)
    if cfg.perceptual_weight > 0:                                          # :402-404
        from . import lpips_oracle
        p_loss = lpips_oracle.distance(lpips_sd, x, dec, dtype).view(-1, 1, 1, 1)
        rec = rec + cfg.perceptual_weight * p_loss
    rec = rec.mean()                                                       # :407
    loss = rec + cfg.codebook_weight * diff                                # :408
    return loss, dict(rec_loss=rec, quant_loss=diff, p_loss=p_loss.mean(), z=z, ind=ind)


def trainable(sd):
    return [k for k in sd if k not in BUFFERS]


def gradients(sd_np, cfg, x, lpips_sd=None):
    """fp64 autograd gradients w.r.t. every trainable tensor -> (grads, metrics)"""
    sd = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=(k not in BUFFERS)) for k, v in sd_np.items()}
    loss, m = losses(sd, cfg, torch.as_tensor(x), lpips_sd=lpips_sd)
    loss.backward()
    grads = {k: (sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])) for k in trainable(sd)}
    return grads, dict(loss=float(loss.detach()), rec_loss=float(m['rec_loss'].detach()), quant_loss=float(m['quant_loss'].detach())), m


def adam_step(params, grads, m, v, step, lr, b1=0.5, b2=0.9, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad), in place on dicts of float64 numpy arrays; ``step`` counts from 1"""
    for k in params:
        g = np.asarray(grads[k], dtype=np.float64)
        m[k] = b1 * m[k] + (1 - b1) * g
        v[k] = b2 * v[k] + (1 - b2) * g * g
        mhat = m[k] / (1 - b1 ** step)
        vhat = v[k] / (1 - b2 ** step)
        params[k] = params[k] - lr * mhat / (np.sqrt(vhat) + eps)
