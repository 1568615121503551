"""Multi-context evaluation loop on MI355X: novel views and camera estimates for EVERY context size in
one transformer pass.

Drop-in for ``generate_batch_predictions`` of viewformer/evaluate/evaluate_transformer_multictx.py:36-95
(same argument meaning, same result keys): the MASK stream (``output_poses`` = the target pose tiled over all
positions) yields the target view generated from 0, 1, ..., S-1 context views, the LOC stream
(``localization_tokens`` = the target view's codes tiled) yields its camera from each context size.
"""
import torch

from . import geometry
from . import ops
from .evaluate import _frames_for_encode


def generate_batch_predictions(transformer_model, codebook_model, images, cameras):
    dev = codebook_model.device
    images = torch.as_tensor(images).to(dev)
    cameras = torch.as_tensor(cameras, dtype=torch.float32).to(dev)
    ground_truth_cameras = cameras[:, -1]
    transform = None
    if transformer_model.config.augment_poses == 'relative':            # :38-40
        cameras, transform = geometry.to_relative_cameras(cameras)
    cameras = geometry.normalize_cameras(cameras)                       # :41

    B, S = images.shape[:2]
    t = transformer_model.config.token_image_size
    codes = codebook_model.encode(_frames_for_encode(images, codebook_model.config.image_size))[-1]     # resize_tf, :45-47
    codes = codes.to(torch.int32).view(B, S, t, t)                      # :53-56

    input_ids = torch.cat([codes[:, :-1], torch.full_like(codes[:, :1], transformer_model.mask_token)], 1)   # :61-62
    context_cameras = torch.cat([cameras[:, :-1], torch.zeros_like(cameras[:, :1])], 1)                      # :63
    query_cameras = cameras[:, -1:].expand(B, S, 7).contiguous()                                               # :66
    query_tokens = codes[:, -1:].expand(B, S, t, t).contiguous()                                               # :67
    output = transformer_model(dict(input_ids=input_ids, poses=context_cameras,
                                    localization_tokens=query_tokens, output_poses=query_cameras), training=False)   # :70-73
    lg = output['logits']                                               # [B,S,t,t,nE]
    nE = lg.shape[-1]
    generated_codes = ops.argmax_rows(lg.reshape(-1, nE), B * S * t * t, nE).view(B * S, t, t)       # :76
    generated_cameras = transformer_model.reduce_cameras(output['pose_prediction'], -2)              # :77  [B,S,7]

    dec = codebook_model.decode_code(generated_codes)                   # :82
    if codebook_model.data_format == 'NCHW':
        dec = dec.permute(0, 2, 3, 1)
    generated_images = ops.postprocess_u8(dec.contiguous())            # :83-84
    generated_images = generated_images.view(B, S, *generated_images.shape[1:])                      # :85

    if transformer_model.config.augment_poses == 'relative':            # :88-89
        generated_cameras = geometry.from_relative_cameras(generated_cameras, transform)
    return dict(ground_truth_images=images[:, -1], generated_images=generated_images,
                ground_truth_cameras=ground_truth_cameras, generated_cameras=generated_cameras,
                generated_codes=generated_codes.view(B, S, t, t), codes=codes)
