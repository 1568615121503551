"""Training-mode codebook quantizer: ``QuantizeEMA.forward`` with ``self.training`` (viewformer/models/utils_th.py:32-68) — the
lookup against the current codebook, then the EMA codebook update with its two replica all-reduces.

The arithmetic is in libvf_hip.so (``vf_vq_argmin_f32``, ``vf_vq_ema_accumulate_f32``, ``vf_vq_ema_update_f32``); torch holds the
buffers and issues the collectives (RCCL on GPUs).  The convolutional encoder/decoder backward of codebook training is not built;
this is the quantizer's own state update, which is what makes the codebook move."""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib, ops
from ._lib import check
from .ops import _f32, _p, _stream


class QuantizeEMATrainer:
    def __init__(self, embeddings, decay: float = 0.99, eps: float = 1e-5, process_group=None,
                 ema_cluster_size_hidden=None, ema_dw_hidden=None, counter: int = 0):
        """``embeddings`` [D, Kc] (the reference's buffer layout, utils_th.py:17-18) on the GPU"""
        E = _f32(embeddings).contiguous().clone()
        if E.device.type != 'cuda':
            raise _lib.VfError('QuantizeEMATrainer runs on the GPU only (no CPU fallback)')
        self.D, self.Kc = E.shape
        self.decay, self.eps, self.group = float(decay), float(eps), process_group
        self.embeddings = E
        self.ema_cluster_size_hidden = (torch.zeros(self.Kc, device=E.device) if ema_cluster_size_hidden is None
                                        else _f32(ema_cluster_size_hidden).contiguous().clone())
        self.ema_dw_hidden = torch.zeros_like(E) if ema_dw_hidden is None else _f32(ema_dw_hidden).contiguous().clone()
        self.counter = int(counter)
        self.training = True
        self._repack()

    def _repack(self):
        self._Ep, self._esq = ops.vq_pack_codebook(self.embeddings)

    def state_dict(self):
        """the reference's buffer names (vqgan_th.py state dict: ``quantize.*``)"""
        return {'quantize.embeddings': self.embeddings.clone(), 'quantize.ema_cluster_size_hidden': self.ema_cluster_size_hidden.clone(),
                'quantize.ema_dw_hidden': self.ema_dw_hidden.clone(), 'quantize.counter': torch.tensor(self.counter, dtype=torch.int64)}

    def __call__(self, z_nchw):
        """-> (quantize [N,D,h,w] straight-through value, diff, embed_ind [N,h,w] int64); updates the codebook when training"""
        lib = _lib.load()
        n, D, h, w = z_nchw.shape
        z = _f32(z_nchw).permute(0, 2, 3, 1).contiguous().view(n * h * w, D)              # utils_th.py:34-35
        M = z.shape[0]
        ind = ops.vq_argmin(z, self._Ep, self._esq, self.D, self.Kc)                      # :36-41 against the CURRENT codebook
        q = ops.codebook_gather(self.embeddings, ind, self.D, self.Kc)                    # :44
        if self.training:
            counts = torch.empty(self.Kc, dtype=torch.float32, device=z.device)
            embed_sum = torch.empty((self.D, self.Kc), dtype=torch.float32, device=z.device)
            check(lib.vf_vq_ema_accumulate_f32(_p(z), _p(ind), M, self.D, self.Kc, _p(counts), _p(embed_sum), _stream()),
                  'vf_vq_ema_accumulate_f32')
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
                dist.all_reduce(counts, group=self.group)                                 # :51
                dist.all_reduce(embed_sum, group=self.group)                              # :52
            self.counter += 1
            corr = float(np.float32(1.0) - np.power(np.float32(self.decay), np.float32(self.counter)))     # :24-30, fp32 like torch
            check(lib.vf_vq_ema_update_f32(_p(counts), _p(embed_sum), _p(self.ema_cluster_size_hidden), _p(self.ema_dw_hidden),
                                           _p(self.embeddings), self.D, self.Kc, self.decay, self.eps, corr, _stream()),
                  'vf_vq_ema_update_f32')
            self._repack()
        diff = (q - z).pow(2).mean()                                                      # :66 (reporting value)
        quant = (z + (q - z)).view(n, h, w, D).permute(0, 3, 1, 2)                        # :67 straight-through value
        return quant, diff, ind.view(n, h, w)
