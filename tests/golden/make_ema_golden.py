#!/usr/bin/env python
"""Golden vectors for the TRAINING branch of the codebook quantizer (QuantizeEMA.forward with self.training,
viewformer/models/utils_th.py:32-68), FROM THE REFERENCE ITSELF.  Run in the build container only:
    python tests/golden/make_ema_golden.py
Three consecutive training-mode calls on seeded inputs; after each call the reference's buffers (embeddings,
ema_cluster_size_hidden, ema_dw_hidden, counter) and outputs (indices, diff) are recorded.  Only data is written."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference      # noqa: E402  (installs the stub modules, puts /root/reference on the path)


def main():
    import_reference()
    from viewformer.models.utils_th import QuantizeEMA
    torch.manual_seed(0)
    D, K = 32, 64
    q = QuantizeEMA(D, K, decay=0.99, eps=1e-5)
    g = np.random.Generator(np.random.PCG64(11))
    E0 = (g.standard_normal((D, K)) * 0.5).astype(np.float32)
    q.embeddings.copy_(torch.from_numpy(E0))
    q.train()
    out = dict(E0=E0, decay=np.float32(0.99), eps=np.float32(1e-5))
    for step in range(3):
        z = (g.standard_normal((3, D, 4, 4)) * 0.6 + 0.05 * step).astype(np.float32)       # [N, D, h, w] like the encoder's output
        quant, diff, ind = q(torch.from_numpy(z))
        out[f'z{step}'] = z
        out[f'ind{step}'] = ind.numpy()
        out[f'diff{step}'] = np.float32(diff.item())
        out[f'quant{step}'] = quant.detach().numpy()
        out[f'E{step + 1}'] = q.embeddings.numpy().copy()
        out[f'cs{step + 1}'] = q.ema_cluster_size_hidden.numpy().copy()
        out[f'dw{step + 1}'] = q.ema_dw_hidden.numpy().copy()
        out[f'counter{step + 1}'] = np.int64(q.counter.item())
    np.savez_compressed(os.path.join(HERE, 'vq_ema.npz'), **out)
    print({k: getattr(v, 'shape', v) for k, v in out.items() if k.startswith(('E', 'cs', 'counter'))})


if __name__ == '__main__':
    main()
