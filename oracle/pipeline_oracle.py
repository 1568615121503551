"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the
evaluator's per-batch hot loop, ``generate_batch_predictions``
(viewformer/evaluate/evaluate_transformer.py:97-146) and of the codebook-only
variant (viewformer/evaluate/evaluate_codebook.py:67-77).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Pinning status: the VQGAN half is pinned by the
golden vectors made from the reference itself; the transformer half is
"parity unpinned" (see migt_oracle.py).
"""
import time

import torch

from . import vqgan_oracle as vq
from . import migt_oracle as mg


def generate_batch_predictions(migt_sd, migt_cfg, vq_sd, vq_cfg, images_u8, cameras, dtype=torch.float32,
                               return_intermediates=False, timings=None):
    """images_u8 [B,S,H,W,3] uint8, cameras [B,S,7] float32 (torch CPU tensors).  ``timings``: a dict that receives the wall
    seconds of the three stages ('encode', 'transformer' = both passes, 'decode') for bench.py's CPU baseline."""
    tm = timings if timings is not None else {}
    t0 = time.perf_counter()

    def lap(key):
        nonlocal t0
        t1 = time.perf_counter()
        tm[key] = tm.get(key, 0.0) + (t1 - t0)
        t0 = t1
    images_u8 = torch.as_tensor(images_u8)
    cameras = torch.as_tensor(cameras, dtype=torch.float32)
    gt_cameras = cameras[:, -1]
    transform = None
    if migt_cfg.augment_poses == 'relative':                    # :99-101
        cameras, transform = mg.to_relative_cameras(cameras)
    cameras = mg.normalize_cameras(cameras)                     # :102

    B, S = images_u8.shape[:2]
    x = vq.preprocess_u8(images_u8.reshape(B * S, *images_u8.shape[2:]))    # :105-108
    codes = vq.encode(vq_sd, vq_cfg, x, dtype)[-1]              # :109
    t = migt_cfg.token_image_size
    codes = codes.to(torch.int32).reshape(B, S, t, t)           # :110,116
    lap('encode')

    ids = torch.cat([codes[:, :-1], torch.full_like(codes[:, :1], migt_cfg.n_embeddings)], 1)   # :120-121
    out = mg.migt_forward(migt_sd, migt_cfg, ids, cameras, dtype=dtype)      # :122
    gen_codes = out['logits'].argmax(-1)[:, -1]                 # :123 (ties -> lowest index by our contract)
    lap('transformer')

    dec = vq.decode_code(vq_sd, vq_cfg, gen_codes, dtype)       # :127
    gen_images = vq.postprocess_u8(dec)                         # :128-129
    lap('decode')

    if migt_cfg.use_localization:                               # :134-136
        out2 = mg.migt_forward(migt_sd, migt_cfg, codes, cameras[:, :-1], dtype=dtype)
        gen_cam = mg.reduce_cameras(out2['pose_prediction'][:, -1:].to(torch.float32), -2)
    else:
        gen_cam = cameras[:, :1]                                # :138
    lap('transformer')
    if migt_cfg.augment_poses == 'relative':                    # :139-140
        gen_cam = mg.from_relative_cameras(gen_cam, transform)
    res = dict(ground_truth_images=images_u8[:, -1], generated_images=gen_images,
               ground_truth_cameras=gt_cameras, generated_cameras=gen_cam[:, -1])
    if return_intermediates:
        res.update(codes=codes, generated_codes=gen_codes, logits_last=out['logits'][:, -1], decoded=dec,
                   cameras=cameras)
    return res


def codebook_batch_predictions(vq_sd, vq_cfg, images_u8, dtype=torch.float32):
    """evaluate_codebook.py:67-77 — encode -> decode round trip (BASELINE config #1)."""
    images_u8 = torch.as_tensor(images_u8)
    x = vq.preprocess_u8(images_u8)
    codes = vq.encode(vq_sd, vq_cfg, x, dtype)[-1]
    dec = vq.decode_code(vq_sd, vq_cfg, codes, dtype)
    return dict(ground_truth_images=images_u8, generated_images=vq.postprocess_u8(dec), codes=codes)
