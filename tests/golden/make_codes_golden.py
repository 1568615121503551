#!/usr/bin/env python
"""Record token indices of MANY full-size images FROM THE REFERENCE ITSELF (flip-rate golden).

Run in the build container only (needs /root/reference):   python tests/golden/make_codes_golden.py

``vqgan_full.npz`` holds 4 images = 256 tokens; a summation-order flip of the fp32 arg-min is a ~1e-4-per-token event, invisible
at that size.  This script runs the reference's own ``VQGAN.encode`` (viewformer/models/vqgan_th.py:379-383, the Torch model the
evaluators load) on 320 synthetic 128x128 frames = 20 480 tokens and records, per token: the reference's code, the runner-up code
and the top-2 margin of the reference's own fp32 ``-dist`` (utils_th.py:36-41).  Only data is written; inputs are regenerated from
the recorded seed by ``viewformer_amd.weights.synthetic_scene_batch``.  The GPU test reports, per arithmetic arm, every token whose
index differs together with its margin (tests/test_hip_models.py::test_token_flip_rate_on_20k_reference_tokens).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, build_reference, CODEBOOK_SCALE                     # noqa: E402
from viewformer_amd.weights import synthetic_scene_batch                                        # noqa: E402

N_SCENES, N_VIEWS, INPUT_SEED, WEIGHT_SEED = 40, 8, 41, 0


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    AutoModelTH, RefCfg = import_reference()
    ref, cfg, _ = build_reference(AutoModelTH, RefCfg, {}, seed=WEIGHT_SEED)
    frames, _ = synthetic_scene_batch(N_SCENES, N_VIEWS, 128, seed=INPUT_SEED)
    frames = frames.reshape(-1, 128, 128, 3)
    q = ref.quantize
    codes, runner, margin = [], [], []
    with torch.no_grad():
        for i in range(0, frames.shape[0], 16):
            x = (torch.from_numpy(frames[i:i + 16]).float() * torch.tensor(1.0 / 255) * 2 - 1).permute(0, 3, 1, 2).contiguous()
            z = ref.quant_conv(ref.encoder(x))
            c = ref.encode(x)[-1]
            f = z.permute(0, 2, 3, 1).reshape(-1, z.size(1))
            dist = f.pow(2).sum(1, keepdim=True) - 2 * f @ q.embeddings + q.embeddings.pow(2).sum(0, keepdim=True)   # utils_th.py:36-40
            t = torch.topk(-dist, 2, dim=1)
            assert torch.equal(t.indices[:, 0].reshape(c.shape), c) or (t.values[:, 0] == t.values[:, 1]).any()
            codes.append(c.numpy().astype(np.int16))
            runner.append(t.indices[:, 1].reshape(c.shape).numpy().astype(np.int16))
            margin.append((t.values[:, 0] - t.values[:, 1]).reshape(c.shape).numpy().astype(np.float32))
            print(i, flush=True)
    codes, runner, margin = np.concatenate(codes), np.concatenate(runner), np.concatenate(margin)
    np.savez_compressed(os.path.join(HERE, 'vqgan_codes_20k.npz'), seed=WEIGHT_SEED, input_seed=INPUT_SEED, n_scenes=N_SCENES,
                        n_views=N_VIEWS, codebook_scale=CODEBOOK_SCALE, codes=codes, runner_up=runner, margin=margin)
    print('tokens', codes.size, 'distinct codes', len(np.unique(codes)), 'median margin', float(np.median(margin)),
          'min margin', float(margin.min()), 'margins < 1e-4:', int((margin < 1e-4).sum()))


if __name__ == '__main__':
    main()
