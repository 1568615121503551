"""The evaluator's per-batch hot loop on MI355X.

Drop-in for ``generate_batch_predictions(transformer_model, codebook_model, images, cameras)``
(viewformer/evaluate/evaluate_transformer.py:97-146) and the codebook-only variant
(viewformer/evaluate/evaluate_codebook.py:67-77): same argument meaning, same result keys.
``images`` uint8 [B,S,H,W,3], ``cameras`` float32 [B,S,7]; tensors may be on the host or on the GPU.
"""
import torch

from . import geometry
from . import ops


def _frames_for_encode(images, image_size):
    """The evaluators resize only INSIDE their ``encode`` helper (resize_tf, evaluate_transformer.py:18-19,105;
    evaluate_transformer_multictx.py:45-47): the caller's frames — and therefore ``ground_truth_images`` (:142) — keep their
    original resolution.  The reference compares ``shape[-2]`` (the width of an NHWC batch) with image_size.
    images [B,S,H,W,3] -> [B*S,h,w,3] for the codebook."""
    B, S = images.shape[:2]
    flat = images.reshape(B * S, *images.shape[2:])
    if images.shape[-2] != image_size and images.shape[-3] != image_size:
        if images.dtype != torch.uint8:
            raise TypeError('frames of another size than config.image_size must be uint8 (data/_common.py:19-45 resizes uint8)')
        flat = ops.resize_u8(flat, image_size)
    return flat


def generate_batch_predictions(transformer_model, codebook_model, images, cameras, return_codes: bool = False,
                               fused_passes: bool = True):
    """``fused_passes``: run the generation pass and the localization pass as one twin-view pass
    (MIGT.generate_and_localize; bit-identical rows, 8/14 of the transformer work at S=7); False = the
    reference's two separate calls."""
    dev = codebook_model.device
    images = torch.as_tensor(images).to(dev)
    cameras = torch.as_tensor(cameras, dtype=torch.float32).to(dev)
    ground_truth_cameras = cameras[:, -1]
    transform = None
    if transformer_model.config.augment_poses == 'relative':            # :99-101
        cameras, transform = geometry.to_relative_cameras(cameras)
    cameras = geometry.normalize_cameras(cameras)                       # :102

    B, S = images.shape[:2]
    t = transformer_model.config.token_image_size
    frames = _frames_for_encode(images, codebook_model.config.image_size)       # resize_tf inside encode(), :105
    # encode every view, target included, exactly as the reference does (:114-116)
    codes = codebook_model.encode(frames)[-1]
    codes = codes.to(torch.int32).view(B, S, t, t)                      # :110,116

    # ``return_codes`` also hands back the last view's logits; without it the arg-max is fused into the LM head's epilogue where the
    # arm supports it (the [B*64, 1024] logits never reach HBM)
    pose_last, lg = None, None
    if transformer_model.use_localization and fused_passes:
        first, pose_last = transformer_model.generate_and_localize(codes, cameras, codes_only=not return_codes)   # :119-123 + :134-136
        if return_codes:
            lg = first
        else:
            generated_codes = first
    else:
        ids = torch.cat([codes[:, :-1], torch.full_like(codes[:, :1], transformer_model.mask_token)], 1)   # :120-121
        out = transformer_model(dict(input_ids=ids, poses=cameras), training=False, last_view_logits_only=return_codes,
                                last_view_codes_only=not return_codes)
        if return_codes:
            lg = out['logits_last']                                     # == output['logits'][:, -1]
        else:
            generated_codes = out['codes_last']
    if lg is not None:
        nE = lg.shape[-1]
        generated_codes = ops.argmax_rows(lg.view(-1, nE), B * t * t, nE).view(B, t, t)   # :123 (ties -> lowest index)
    generated_codes = generated_codes.view(B, t, t)

    dec = codebook_model.decode_code(generated_codes)                   # :127
    if codebook_model.data_format == 'NCHW':
        dec = dec.permute(0, 2, 3, 1)
    generated_images = ops.postprocess_u8(dec.contiguous())             # :128-129

    if transformer_model.use_localization:                              # :134-136
        if pose_last is None:
            out2 = transformer_model(dict(input_ids=codes, poses=cameras[:, :-1]), training=False,
                                     last_view_logits_only=True)
            pose_last = out2['pose_prediction'][:, -1:]
        generated_cameras = transformer_model.reduce_cameras(pose_last, -2)
    else:
        generated_cameras = cameras[:, :1]                              # :138
    if transformer_model.config.augment_poses == 'relative':            # :139-140
        generated_cameras = geometry.from_relative_cameras(generated_cameras, transform)
    res = dict(ground_truth_images=images[:, -1], generated_images=generated_images,
               ground_truth_cameras=ground_truth_cameras, generated_cameras=generated_cameras[:, -1])
    if return_codes:
        res.update(codes=codes, generated_codes=generated_codes, logits_last=lg, decoded=dec, pose_last=pose_last)
    return res


def codebook_batch_predictions(codebook_model, images):
    """evaluate_codebook.py:67-77 — encode -> decode round trip (BASELINE config #1)."""
    dev = codebook_model.device
    images = torch.as_tensor(images).to(dev)
    frames = images
    if images.dtype == torch.uint8:                                       # resize_tf, evaluate_codebook.py:15-16,68 (identity at image_size)
        frames = ops.resize_u8(images, codebook_model.config.image_size)
    codes = codebook_model.encode(frames)[-1]
    dec = codebook_model.decode_code(codes)
    if codebook_model.data_format == 'NCHW':
        dec = dec.permute(0, 2, 3, 1)
    return dict(ground_truth_images=images, generated_images=ops.postprocess_u8(dec.contiguous()), codes=codes)
