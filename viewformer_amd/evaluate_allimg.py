"""All-images evaluation loop on MI355X (BASELINE configs[4]: "large-batch AR decode").

Mirror of viewformer/evaluate/evaluate_transformer_multictx_allimg.py: for one sequence of F frames and a fixed set of context views,
EVERY frame is generated as the target of that context — ``encode_images`` (:66-81) once for the whole sequence,
``transformer_predict`` (:15-48: the multi-context MASK / LOC stream pass) over the F (context + target) scenes in batches of 128
(:173), ``decode_code`` (:84-91) in batches of 64 scenes (:177) — same function names, argument meaning and results.  The
``keep_last_frame`` variant (:147-168, a sequential chain of 1-scene calls) is the same ``transformer_predict`` in a Python loop.
"""
import numpy as np
import torch

from . import geometry
from . import ops
from .evaluate import _frames_for_encode

TRANSFORMER_BATCH = 128          # evaluate_transformer_multictx_allimg.py:173
DECODE_BATCH = 64                # :177


def transformer_predict(cameras, codes, *, transformer_model):
    """:15-48 — codes [N,S,t,t] int, cameras [N,S,7] -> (generated_cameras [N,S,7] or None, generated_codes [N,S,t,t] int64)"""
    transform = None
    if transformer_model.config.augment_poses == 'relative':            # :16-18
        cameras, transform = geometry.to_relative_cameras(cameras)
    cameras = geometry.normalize_cameras(cameras)                       # :19
    N, S = codes.shape[:2]
    t = codes.shape[-1]
    input_ids = torch.cat([codes[:, :-1], torch.full_like(codes[:, :1], transformer_model.mask_token)], 1)   # :25-26
    context_cameras = torch.cat([cameras[:, :-1], torch.zeros_like(cameras[:, :1])], 1)                      # :27
    query_cameras = cameras[:, -1:].expand(N, S, 7).contiguous()                                               # :30
    query_tokens = codes[:, -1:].expand(N, S, t, t).contiguous()                                               # :31
    output = transformer_model(dict(input_ids=input_ids, poses=context_cameras, localization_tokens=query_tokens,
                                    output_poses=query_cameras), training=False)                               # :34-37
    lg = output['logits']
    nE = lg.shape[-1]
    generated_codes = ops.argmax_rows(lg.reshape(-1, nE), N * S * t * t, nE).view(N, S, t, t)                # :40
    generated_cameras = None
    if 'pose_prediction' in output:                                                                            # :41-47
        generated_cameras = transformer_model.reduce_cameras(output['pose_prediction'], -2)
        if transformer_model.config.augment_poses == 'relative':
            generated_cameras = geometry.from_relative_cameras(generated_cameras, transform)
    return generated_cameras, generated_codes


def run_with_batchsize(fn, batch_size, *args, **kwargs):
    """:51-63 — apply ``fn`` to slices of ``batch_size`` along dim 0 and concatenate (a tensor, or a tuple with ``None`` members)"""
    total = len(args[0])
    outs = [fn(*[x[i:i + batch_size] for x in args], **kwargs) for i in range(0, total, batch_size)]
    if torch.is_tensor(outs[0]):
        return torch.cat(outs, 0)
    return tuple(torch.cat([o[i] for o in outs], 0) if outs[0][i] is not None else None for i in range(len(outs[0])))


def encode_images(frames, *, codebook_model):
    """:66-81 — frames [B,S,H,W,3] uint8 -> codes [B,S,t,t] int32 (resize inside, as the reference's ``encode``)"""
    frames = torch.as_tensor(frames).to(codebook_model.device)
    B, S = frames.shape[:2]
    codes = codebook_model.encode(_frames_for_encode(frames, codebook_model.config.image_size))[-1].to(torch.int32)
    return codes.view(B, S, *codes.shape[-2:])


def decode_code(generated_codes, *, codebook_model):
    """:84-91 — codes [N,S,t,t] -> uint8 images [N,S,H,W,3]"""
    N, S = generated_codes.shape[:2]
    dec = codebook_model.decode_code(generated_codes.reshape(N * S, *generated_codes.shape[2:]))
    if codebook_model.data_format == 'NCHW':
        dec = dec.permute(0, 2, 3, 1)
    img = ops.postprocess_u8(dec.contiguous())
    return img.view(N, S, *img.shape[1:])


def evaluate_sequence(transformer_model, codebook_model, frames, cameras, context_views, keep_last_frame: bool = False):
    """The per-sequence body of ``main`` (:128-177).  frames [F,H,W,3] uint8, cameras [F,7]; ``context_views``: indices into the
    sequence.  Returns generated_images [F,S,H,W,3] uint8, generated_cameras [F,S,7] or None, generated_codes [F,S,t,t], codes [F,t,t]
    (S = len(context_views) + 1: position s is the target generated from the first s context views, multi-context semantics)."""
    dev = codebook_model.device
    frames = torch.as_tensor(frames).to(dev)[None]                      # :135 (batch of one sequence)
    cameras = torch.as_tensor(np.asarray(cameras), dtype=torch.float32).to(dev)[None]
    F = frames.shape[1]
    codes = encode_images(frames, codebook_model=codebook_model)        # :138  [1,F,t,t]
    ctx = [int(j) for j in context_views]
    idx = torch.tensor([ctx + [i] for i in range(F)], device=dev)       # :140-141: scene i = (context..., frame i)
    tcodes = codes[0][idx]                                              # [F,S,t,t]
    tcameras = cameras[0][idx]                                          # [F,S,7]
    if keep_last_frame:                                                 # :145-168: each call also sees the previous generated frame
        gen_codes, gen_cams, last = [], [], None
        disable_cameras = False
        for i in range(F):
            lcodes, lcams = tcodes[i:i + 1], tcameras[i:i + 1]
            if last is not None:
                lcodes = torch.cat([last[0].to(lcodes.dtype), lcodes], 1)
                lcams = torch.cat([last[1], lcams], 1)
            lgcams, lgcodes = transformer_predict(lcams, lcodes, transformer_model=transformer_model)
            if last is not None:
                lgcodes = lgcodes[:, 1:]
                if lgcams is not None:
                    lgcams = lgcams[:, 1:]
            gen_codes.append(lgcodes)
            if lgcams is not None:
                gen_cams.append(lgcams)
            else:
                disable_cameras = True
            last = (lgcodes[:, -1:], lcams[:, -1:])
        generated_codes = torch.cat(gen_codes, 0)
        generated_cameras = None if disable_cameras else torch.cat(gen_cams, 0)
    else:
        generated_cameras, generated_codes = run_with_batchsize(transformer_predict, TRANSFORMER_BATCH, tcameras, tcodes,
                                                                transformer_model=transformer_model)           # :173
    generated_images = run_with_batchsize(decode_code, DECODE_BATCH, generated_codes, codebook_model=codebook_model)   # :177
    return dict(generated_images=generated_images, generated_cameras=generated_cameras, generated_codes=generated_codes,
                codes=codes[0], eval_frames=[x for x in range(F) if x not in ctx])                                   # :178
