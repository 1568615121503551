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


CAMERA_SIDE_STREAM = True       # the O(B*S) camera bookkeeping (geometry.py: ~100 element-wise launches of 4-5 us per batch, 0.5 ms of a 113 ms
                                # step when they queue between the model kernels) runs on a second HIP stream beside the encoder / decoder
                                # kernels; same torch operations on the same values — results bit-identical (tests/test_hip_evaluate_loop.py)
_side_streams = {}


def _side_stream(dev):
    key = torch.device(dev).index or 0
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(dev)
    return _side_streams[key]


class _beside:
    """``with _beside(dev, *inputs) as b: ...`` runs the block's launches on the side stream after everything queued on the current stream
    so far (the inputs are ready); ``b.join(*outputs)`` makes the current stream wait for them.  Tensors that cross streams are recorded on
    the stream that did not allocate them (caching-allocator rule)."""

    def __init__(self, dev, *inputs):
        self.on = CAMERA_SIDE_STREAM and torch.device(dev).type == 'cuda'
        if self.on:
            self.cur, self.side = torch.cuda.current_stream(dev), _side_stream(dev)
            self.ctx = torch.cuda.stream(self.side)
            self.inputs = inputs

    def __enter__(self):
        if self.on:
            self.side.wait_stream(self.cur)
            for t in self.inputs:
                if t is not None:
                    t.record_stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *outputs):
        if self.on:
            self.cur.wait_stream(self.side)
            for t in outputs:
                if t is not None:
                    t.record_stream(self.cur)


MAX_SCENES_PER_CALL = 256      # scenes per transformer / decoder pass: the reference's loop batches arbitrary sizes
                               # (evaluate_transformer.py:219); scenes are independent, so a larger batch is walked in chunks with
                               # identical results (1024 scenes = 7168 frames would otherwise need 32-bit-offset-breaking activations)


def generate_batch_predictions(transformer_model, codebook_model, images, cameras, return_codes: bool = False,
                               fused_passes: bool = True, max_scenes_per_call: int = None):
    """``fused_passes``: run the generation pass and the localization pass as one twin-view pass
    (MIGT.generate_and_localize; bit-identical rows, 8/14 of the transformer work at S=7); False = the
    reference's two separate calls.  Batches above ``max_scenes_per_call`` (default MAX_SCENES_PER_CALL) are processed in scene
    chunks and concatenated."""
    dev = codebook_model.device
    images = torch.as_tensor(images).to(dev)
    cameras = torch.as_tensor(cameras, dtype=torch.float32).to(dev)
    chunk = max_scenes_per_call or MAX_SCENES_PER_CALL
    if images.shape[0] > chunk:
        parts = [generate_batch_predictions(transformer_model, codebook_model, images[i:i + chunk], cameras[i:i + chunk], return_codes,
                                            fused_passes, chunk) for i in range(0, images.shape[0], chunk)]
        # (a key that is None for a chunk — pose_last without localization — stays None: the schema does not depend on the batch size)
        return {k: (None if parts[0][k] is None else torch.cat([p[k] for p in parts])) for k in parts[0]}
    ground_truth_cameras = cameras[:, -1]
    transform = None
    with _beside(dev, cameras) as frames_change:                        # (beside the encoder: CAMERA_SIDE_STREAM)
        if transformer_model.config.augment_poses == 'relative':        # :99-101
            cameras, transform = geometry.to_relative_cameras(cameras)
        cameras = geometry.normalize_cameras(cameras)                   # :102

    B, S = images.shape[:2]
    t = transformer_model.config.token_image_size
    frames = _frames_for_encode(images, codebook_model.config.image_size)       # resize_tf inside encode(), :105
    # encode every view, target included, exactly as the reference does (:114-116)
    codes = codebook_model.encode(frames)[-1]
    codes = codes.to(torch.int32).view(B, S, t, t)                      # :110,116

    # ``return_codes`` also hands back the last view's logits; without it the arg-max is fused into the LM head's epilogue where the
    # arm supports it (the [B*64, 1024] logits never reach HBM)
    pose_last, lg = None, None
    frames_change.join(cameras, transform)
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

    def camera_tail(pose_last):
        with _beside(dev, pose_last, cameras, transform) as tail:       # (beside the decoder when the fused pass already has the pose)
            if transformer_model.use_localization:                      # :134-136
                generated_cameras = transformer_model.reduce_cameras(pose_last, -2)
            else:
                generated_cameras = cameras[:, :1]                      # :138
            if transformer_model.config.augment_poses == 'relative':    # :139-140
                generated_cameras = geometry.from_relative_cameras(generated_cameras, transform)
        return tail, generated_cameras

    tail = None
    if pose_last is not None or not transformer_model.use_localization:
        tail, generated_cameras = camera_tail(pose_last)
    dec = codebook_model.decode_code(generated_codes)                   # :127
    if codebook_model.data_format == 'NCHW':
        dec = dec.permute(0, 2, 3, 1)
    generated_images = ops.postprocess_u8(dec.contiguous())             # :128-129
    if tail is None:                                                    # the reference's separate localization call, :134-136
        out2 = transformer_model(dict(input_ids=codes, poses=cameras[:, :-1]), training=False, last_view_logits_only=True)
        pose_last = out2['pose_prediction'][:, -1:]
        tail, generated_cameras = camera_tail(pose_last)
    tail.join(generated_cameras)
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


def stream_batch_predictions(transformer_model, codebook_model, batches, depth: int = 2):
    """The evaluator's OUTER loop (evaluate_transformer.py:219-222: ``for batch in dataset: generate_batch_predictions(...)`` with
    host tensors in and out) with the host round trip hidden: batch i+1's frames / cameras are uploaded and batch i-1's results are
    downloaded on a second HIP stream while batch i computes.  ``batches`` yields ``(images uint8 [B,S,H,W,3], cameras [B,S,7])``
    host tensors (pinned ones copy asynchronously); yields per batch the same dict as ``generate_batch_predictions`` with
    ``generated_images`` / ``generated_cameras`` on the host.  Results are those of the plain call, in order.  The host tensors
    come from a ring of ``2 * depth`` pinned buffers per key: batch j is yielded once download j + depth - 1 has been issued and its
    buffer is reused by download j + 2 * depth, so a yielded batch's host tensors stay valid until ``depth`` further batches have been
    REQUESTED (the reference's loop consumes a batch's images before it asks for the next one); copy what must live longer."""
    dev = codebook_model.device
    compute = torch.cuda.current_stream(dev)
    copy = torch.cuda.Stream(dev)
    ring, ring_pos = {}, [0]

    def host_buffer(k, t):
        key = (k, tuple(t.shape), t.dtype)
        if key not in ring:
            ring[key] = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for _ in range(2 * depth)]
        return ring[key][ring_pos[0] % (2 * depth)]

    def upload(b):
        img, cam = (torch.as_tensor(x) for x in b)
        with torch.cuda.stream(copy):
            img_d = img.to(dev, non_blocking=True)
            cam_d = cam.to(dev, torch.float32, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy)
        return img_d, cam_d, ev

    def download(out):
        done = torch.cuda.Event()
        done.record(compute)
        host = {}
        with torch.cuda.stream(copy):
            copy.wait_event(done)
            for k in ('generated_images', 'generated_cameras'):
                out[k].record_stream(copy)
                host[k] = host_buffer(k, out[k])
                host[k].copy_(out[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy)
        ring_pos[0] += 1
        return out, host, ev

    it = iter(batches)
    pending = []                       # results whose download is in flight
    nxt = next(it, None)
    up = upload(nxt) if nxt is not None else None
    while up is not None:
        img_d, cam_d, ev = up
        nxt = next(it, None)
        up = upload(nxt) if nxt is not None else None          # the next batch's copy overlaps this batch's kernels
        compute.wait_event(ev)
        img_d.record_stream(compute)
        cam_d.record_stream(compute)
        pending.append(download(generate_batch_predictions(transformer_model, codebook_model, img_d, cam_d)))
        while len(pending) >= depth:
            out, host, dl = pending.pop(0)
            dl.synchronize()
            yield dict(out, **host)
    for out, host, dl in pending:
        dl.synchronize()
        yield dict(out, **host)
