"""One data-parallel training step of the MIGT transformer on MI355X (BASELINE config #4).

Host-side mirror of ``MIGT.train_step`` (viewformer/models/migt.py:464-505): multi-stream forward with saved
activations (streams of :392-401 laid out as extra views, attention by the STREAMS mask), the losses of :416-448
(token cross-entropy on the MASK stream, pose MSE on the LOC stream, views >= n_loss_skip, per-sample means,
``reduce_mean`` over the local batch :476), backward, optional per-tensor ``clip_by_norm`` (:486-487), gradient
all-reduce SUM across replicas (what MirroredStrategy does under ``apply_gradients`` :488 — per-replica mean loss,
summed gradients, no division by the world size) and the AdamWeightDecay update with the warm-up + cosine schedule
of ``create_optimizer`` (viewformer/models/utils.py:371-437,440-564; compile() :457-462 uses 2000 warm-up steps).

All arithmetic is in libvf_hip.so: the backward's dense contractions are vf_igemm_f32 launches
(dX = dY.W^T with transposed-packed weights, dW = X^T.dY through a transpose + packed dY), attention backward
re-materialises the probabilities per head with batched GEMMs (first version: dense, no causal skipping), the rest
are the HBM-bound kernels of csrc/train_ops.hip.  Parameters, gradients and Adam moments live in flat fp32
buffers so that a layer's gradients are one contiguous RCCL all-reduce issued as soon as that layer's backward
is done (overlapping the remaining backward).

Dropout (config.dropout, default 0.1 in the reference: migt.py:72,216,403 and attn_dropout branching_attention.py:15-17) uses a
counter-based mask — keep = hash(seed, site, element index) >= rate * 2^32 (csrc/vf_common.h) — that the backward pass recomputes;
``dropout_seed`` and the step counter give the per-step seed.  The masks cannot coincide with TensorFlow's RNG stream; the tests
check the kernels against autograd with the SAME masks restated in numpy (oracle/train_oracle.py).  In the bf16 arm (the timed one: the
reference trains with --fp16) no dropout site is a pass of its own except the embedding's: the residual / MLP masks are applied by the
epilogue of the projection GEMM (igemm(drop=...)), their backward by the LayerNorm backward that writes the bf16 gradient operand
(layernorm_bwd(drop=...)), the attention's inside the flash forward / backward kernels.

``random_pose_multiplier`` (migt.py:350-354,160-161): each scene's positions are scaled by rpm ** u, u ~ U(-1, 1), on the way in
and the predicted position is divided by the same factor before the loss; u comes from the same counter hash as the dropout masks
(site SITE_POSE_MULT, element = scene index), restated in oracle/train_oracle.py.  ``use_dynamic_pose_loss`` (:107-120,279-284,440):
the learned log-variance pair ``pose_loss_weighting_criterion.pos_ori_weights`` — note the reference SUMS this term over the local
batch (reduce_sum, :117) while every other term is a batch mean; restated as is.
"""
import math
import re

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from . import train_ops as T
from . import geometry
from . import sharding
from .migt import MIGT


def parse_schedule(s, total_steps=None):
    """'1', '5.', 'linear(a,b[,N])', 'cosine(a,b[,N])', 'warmup(<inner>,W)' -> callable(step) with ``is_zero()``
    (viewformer/utils/schedules.py:72-248; a missing N is the model's total_steps, migt.py:268)"""
    from .schedules import parse
    sch = parse(s)
    return sch.with_total_steps(total_steps) if total_steps is not None else sch


def learning_rate(step, init_lr, total_steps, warmup_steps, offset: int = 0):
    """WarmUp(CosineDecay) of create_optimizer (models/utils.py:310-361,403-412); ``step`` = optimizer.iterations; ``offset`` = WarmUp.offset
    (models/utils.py:337,341: the schedule counts from there — set by the finetune driver to the restored iteration count)"""
    step = max(step - offset, 0)
    if warmup_steps and step < warmup_steps:
        return init_lr * (step / warmup_steps)
    decay_steps = max(total_steps - warmup_steps, 1)
    t = min(step - warmup_steps, decay_steps)
    return init_lr * 0.5 * (1.0 + math.cos(math.pi * t / decay_steps))


def pose_augmentation_draws(augment: str, n_scenes: int, generator=None, dtype=torch.float32):
    """the random numbers one ``process_batch`` call per scene consumes (train_transformer.py:39-52): a dict of per-scene tensors —
    'shift' [n,3] ~ N(0,1); 'y0' / 'y1' [n] ~ U(0, 2 pi) and 'x' [n] ~ U(0, pi / 8) for 'simple'; 'y0' only for 'advanced'.  The reference draws
    from TensorFlow's global generator, so the stream itself cannot be reproduced — the distributions and their use are what is mirrored."""
    if augment not in ('simple', 'advanced'):
        return {}
    def uni(hi):
        return torch.rand((n_scenes,), generator=generator, dtype=dtype) * hi
    d = dict(shift=torch.randn((n_scenes, 3), generator=generator, dtype=dtype), y0=uni(2 * math.pi))
    if augment == 'simple':
        d.update(x=uni(math.pi / 8), y1=uni(2 * math.pi))
    return d


def process_batch(cameras, tokens, augment: str, split: str, draws=None, generator=None):
    """The token dataset's per-sequence transform (viewformer/train/train_transformer.py:28-61, mapped over every sampled sequence by
    load_token_dataset, data/tfrecord_dataset.py:177-184): ``cameras`` [S,7] or [B,S,7] (xyz + quaternion w,x,y,z) -> the poses the
    model trains on.  'relative': the first view becomes the frame (:31-36); 'no', or any mode outside the training split: unchanged
    (:37-38); 'simple': a random shift and a random rotation Ry * (Rx * Ry) of the whole scene (:39-50); 'advanced': shift + Ry (:51-55);
    then the quaternions are normalised and sign-fixed (:60-61).  One set of random numbers per sequence, as in the reference (its
    shapes are (1, 3) / (1,): broadcast over the sequence's views).  ``draws`` = pose_augmentation_draws(...) to make a call reproducible."""
    cameras = torch.as_tensor(cameras)
    single = cameras.dim() == 2
    cam = cameras[None] if single else cameras
    xyz, quaternion = cam[..., :3], cam[..., 3:]
    if augment == 'relative':
        rotation_inverse = geometry.quaternion_conjugate(quaternion[..., :1, :])
        xyz = geometry.quaternion_rotate(xyz - xyz[..., :1, :], rotation_inverse)
        quaternion = geometry.quaternion_multiply(rotation_inverse, quaternion)
    elif augment == 'no' or split != 'train':
        pass
    elif augment in ('simple', 'advanced'):
        d = draws if draws is not None else pose_augmentation_draws(augment, cam.shape[0], generator, cam.dtype)
        d = {k: v.to(cam.device, cam.dtype) for k, v in d.items()}
        xyz = xyz + d['shift'][:, None, :]
        rotation = geometry.make_quaternion_y(d['y0'])
        if augment == 'simple':
            rotation = geometry.quaternion_multiply(rotation, geometry.quaternion_multiply(geometry.make_quaternion_x(d['x']),
                                                                                            geometry.make_quaternion_y(d['y1'])))
        rotation = rotation[:, None, :]
        xyz = geometry.quaternion_rotate(xyz, rotation)
        quaternion = geometry.quaternion_multiply(quaternion, rotation)
    else:
        raise ValueError(f'Augment {augment} is not supported')
    quaternion = geometry.quaternion_remove_sign(geometry.quaternion_normalize(quaternion))
    out = torch.cat([xyz, quaternion], -1)
    return (out[0] if single else out), tokens


# dropout sites (the `site` word of the counter-based mask): one per Dropout layer instance of the reference
SITE_EMBED = 1
SITE_POSE_MULT = 2                      # per-scene random pose multiplier (host-side draw)
DYN_KEY = 'pose_loss_weighting_criterion.pos_ori_weights'


def site_attn(i):
    return 16 + 4 * i


def site_resid(i):
    return 17 + 4 * i


def site_mlp(i):
    return 18 + 4 * i


class MIGTTrainer:
    def __init__(self, model: MIGT, warmup_steps: int = 2000, beta1: float = 0.9, beta2: float = 0.999,
                 eps: float = 1e-8, process_group=None):
        cfg = model.config
        if not 0.0 <= cfg.dropout < 1.0:
            raise ValueError('dropout must be in [0, 1)')
        self.dropout_seed = 0
        if not cfg.random_pose_multiplier > 0:
            raise ValueError('random_pose_multiplier must be positive')
        if cfg.use_dynamic_pose_loss and DYN_KEY not in (model._sd_host or {}):
            raise RuntimeError(f'Missing keys: {DYN_KEY} (use_dynamic_pose_loss)')
        if model._sd_host is None or model.device is None:
            raise RuntimeError('load_state_dict() and .to("cuda") the model first')
        self.model, self.cfg, self.dev = model, cfg, model.device
        self.warmup_steps, self.b1, self.b2, self.eps = warmup_steps, beta1, beta2, eps
        self.group = process_group
        self.step_count = 0                          # optimizer.iterations == model._train_counter
        self.lr_offset = 0                           # WarmUp.offset (models/utils.py:337): begin_finetune() moves it
        self.lr_init, self.lr_total_steps = cfg.learning_rate, cfg.total_steps     # create_optimizer's arguments (train_transformer.py)
        self.loc_weight = parse_schedule(cfg.localization_weight, cfg.total_steps)
        self._layout()
        self._bind()

    # ------------------------------------------------------------------ flat parameter storage
    def _layout(self):
        h, c = self.model._sd_host, self.cfg
        head = ['wte.weight', 'wpe.embeddings']
        for n in ('pose_embedding.c_fc', 'pose_embedding.c_proj', 'pose_criterion.pose_classifier.c_fc',
                  'pose_criterion.pose_classifier.c_proj'):
            head += [n + '.weight', n + '.bias']
        head += ['ln_f.gamma', 'ln_f.beta']
        if c.use_dynamic_pose_loss and self.model.use_localization:
            head.append(DYN_KEY)
        layers = []
        for i in range(c.n_layer):
            p = f'h.{i}'
            names = [p + '.ln_1.gamma', p + '.ln_1.beta', p + '.attn.c_attn.weight', p + '.attn.c_attn.bias',
                     p + '.attn.c_proj.weight', p + '.attn.c_proj.bias', p + '.ln_2.gamma', p + '.ln_2.beta',
                     p + '.mlp.c_fc.weight', p + '.mlp.c_fc.bias', p + '.mlp.c_proj.weight', p + '.mlp.c_proj.bias']
            layers.append(names)
        self.names = head + [n for l in layers for n in l]
        self.slices, off = {}, 0
        for n in self.names:
            shape = tuple(h[n].shape)
            k = int(np.prod(shape))
            self.slices[n] = (off, off + k, shape)
            off += (k + 127) // 128 * 128                # every tensor starts on a 512-byte boundary (float4 kernels need 16; the fused optimizer's
                                                         # 16 x 128 tiles write whole 128-byte lines only when a matrix row starts on one — with the
                                                         # 7-wide pose tensors in front, 16-byte alignment left every tile edge a shared line)
        self.total = off
        self.head_range = (0, self.slices[layers[0][0]][0] if layers else off)
        self.layer_ranges = [(self.slices[l[0]][0], self.slices[l[-1]][1]) for l in layers]
        dev = self.dev
        self.flat_p = torch.zeros(self.total, dtype=torch.float32, device=dev)
        for n in self.names:
            a, b, _ = self.slices[n]
            self.flat_p[a:b] = torch.from_numpy(h[n].reshape(-1)).to(dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        self._scratch = torch.zeros(1, dtype=torch.float32, device=dev)

    def p(self, name):
        a, b, s = self.slices[name]
        return self.flat_p[a:b].view(s)

    def g(self, name):
        a, b, s = self.slices[name]
        return self.flat_g[a:b].view(s)

    def _bind(self):
        """point the model's tensors at the flat buffer and (re)build every packed weight"""
        m, c = self.model, self.cfg
        m._wte, m._wpe = self.p('wte.weight'), self.p('wpe.embeddings')
        for name, dn in m._dense.items():
            dn.w_raw, dn.bias = self.p(name + '.weight'), self.p(name + '.bias')
        for name in list(m._ln):
            m._ln[name] = (self.p(name + '.gamma'), self.p(name + '.beta'))
        self.wpT = getattr(self, 'wpT', {})
        self.repack()

    def repack(self):
        """refresh every packed form of the (just updated) weights: native-f32 packings always (odd shapes, the LM head, the
        pose heads), and with ``dense_arith='x6'`` the fp32-equivalent split packings the dense layers actually run on —
        forward W, and W^T for dX (x6 GEMMs are ~1.8x the native ones; the packers are a few launches per layer)"""
        m, c = self.model, self.cfg
        nE, d = c.n_embeddings, c.d_model
        bf16 = m.precision == 'bf16'          # the reference trains with --fp16 (mixed_float16): bf16 MFMA, fp32 master weights
        x6 = m.dense_arith in ('x6', 'x3h') and not bf16
        x3h = m.dense_arith == 'x3h'          # forward activations only; dX / dW (gradient magnitudes) stay on x6
        self.wpT6 = getattr(self, 'wpT6', {})
        self.wpT16 = getattr(self, 'wpT16', {})
        m._lm_head16 = None
        m._lm_head6 = None                                  # the LM head / pose heads stay on the native kernel in training
        # bf16 arm: after the first call (which allocates the packings one by one) every layer's W and W^T packing is refreshed by ONE
        # launch over a descriptor table (96 pack launches per step before)
        lm16 = bf16 and self.bf16_lm_head and d % 256 == 0 and nE % 256 == 0
        if lm16 and self._lm16 is None:
            self._pack16 = None                                  # (the LM-head packings join the descriptor table: rebuild it)
            self._adam_pack = None
        if self._pack16 is not None and (not bf16 or any(
                dn.wp16 is None or name not in self.wpT16 for name, dn in m._dense.items()
                if dn.k % 128 == 0 and dn.n % 128 == 0)):
            # the table holds raw pointers into packings that are gone (the arm was switched to f32 and back, or a layer lost its
            # buffers): never run it against them — rebuild from scratch below
            self._pack16 = None
            self._pack16_keep = None
            self._layer_pack_ranges = None
            self._adam_pack = None
        packs_fresh, self._packs_fresh = self._packs_fresh, False
        if self._pack16 is not None and not packs_fresh:          # (fresh: apply_gradients' fused optimizer wrote them with the update)
            if self._early_layers_done and self._layer_pack_ranges is not None:
                # the layers' packings were refreshed right after their early update (train_step): only the rest of the table here
                lo = min(a for a, n_ in self._layer_pack_ranges)
                hi = max(a + n_ for a, n_ in self._layer_pack_ranges)
                self._pack16(0, lo)
                self._pack16(hi, None)
            else:
                self._pack16()
        pack_items, pack_names = [], []
        for name, dn in m._dense.items():
            k32 = dn.k % 32 == 0
            to16 = bf16 and k32 and dn.k % 128 == 0 and dn.n % 128 == 0
            to6 = x6 and k32 and dn.k % 64 == 0 and dn.n % 64 == 0
            # the native-f32 packings are refreshed only where no faster arm applies (they were 477 pack launches per step that nothing
            # read: the wide layers run on their bf16 / split packings); the transposed native packing of a wide layer is built on demand
            # in _linear_bwd for row counts its arm does not tile
            dn.wp = ops.pack(dn.w_raw, dn.k, dn.n, 1, sk=dn.n, sn=1, st=0, out=dn.wp) if k32 and not (to16 or to6) else None    # forward: x @ W
            if dn.n % 32 == 0 and not (to16 or to6):                                                          # dX = dY @ W^T
                self.wpT[name] = ops.pack(dn.w_raw, dn.n, dn.k, 1, sk=1, sn=dn.n, st=0, out=self.wpT.get(name))
            else:
                self.wpT.pop(name, None)
            keep16 = dn.wp16 if (to16 and self._pack16 is not None) else None
            dn.wp6 = dn.wp16 = None
            if to16:
                if self._pack16 is not None:                          # buffers of the one-launch refresh below: filled there
                    dn.wp16 = keep16
                else:
                    dn.wp16 = ops.pack_dense_kn_bf16(dn.w_raw)
                    self.wpT16[name] = ops.pack_dense_nk_bf16(dn.w_raw)
                    pack_items += [(dn.w_raw, False, dn.wp16), (dn.w_raw, True, self.wpT16[name])]
                    pack_names += [name, name]
            if to6:
                dn.wp6 = ops.pack_dense_kn_x3h(dn.w_raw) if x3h else ops.pack_dense_kn_x6(dn.w_raw)
                self.wpT6[name] = ops.pack_dense_nk_x6(dn.w_raw)          # [K][N] read as the transposed [N][K] operand
        if lm16 and self._lm16 is None:                              # tied LM head on the bf16 pipe: logits = h @ wte^T, dH = dlogits @ wte
            head = m._wte[:nE]
            self._lm16 = (ops.pack_dense_nk_bf16(head), ops.pack_dense_kn_bf16(head))
            pack_items += [(head, True, self._lm16[0]), (head, False, self._lm16[1])]
            pack_names += ['wte', 'wte']
        elif lm16 and self._pack16 is None:                          # (per-tensor refresh: one_launch_repack off)
            head = m._wte[:nE]
            self._lm16 = (ops.pack_dense_nk_bf16(head), ops.pack_dense_kn_bf16(head))
        if bf16 and self._pack16 is None and pack_items and self.one_launch_repack:
            self._pack16 = ops.pack_bf16_multi(pack_items)          # (re-packs once more; from now on the closure is the refresh)
            self._pack16_keep = pack_items                          # (the closure holds raw pointers: keep the tensors alive with it)
            # the same packings as destinations of the fused optimizer (apply_gradients): one descriptor per weight matrix with both of its
            # packed forms; None when a packing's source is not a whole [rows][cols] view of the flat buffer or a shape does not tile
            by_src, ok = {}, True
            for w, tr, out in pack_items:
                e = by_src.setdefault(w.data_ptr(), [w, None, None])
                ok = ok and e[0].shape == w.shape and e[2 if tr else 1] is None
                e[2 if tr else 1] = out
            if self._nodecay is None:
                r = [[self.slices[n][0], self.slices[n][1]] for n in self.names if 'bias' in n]
                self._nodecay = torch.tensor(sorted(r), dtype=torch.int64, device=self.dev).reshape(-1, 2)
            self._adam_pack = T.adamw_pack_table(self.flat_p, [tuple(e) for e in by_src.values()], self._nodecay) if ok else None
            # descriptor ranges per transformer layer (early_optimizer: a layer's weights are re-packed as soon as they are updated); valid only
            # when every layer's descriptors are contiguous and every wide layer weight is in the table
            rng, ok = [], True
            for i in range(c.n_layer):
                idx = [j for j, nm in enumerate(pack_names) if nm.startswith(f'h.{i}.')]
                ok = ok and len(idx) == 8 and idx == list(range(idx[0], idx[0] + 8))
                rng.append((idx[0], len(idx)) if idx else (0, 0))
            self._layer_pack_ranges = rng if ok else None
        m._lm_head = ops.pack(m._wte, d, nE, 1, sk=1, sn=d, st=0, out=m._lm_head)                          # logits = h @ wte^T (kept fresh for
        if not lm16:                                                                                      # whoever reads the model afterwards)
            self.lm_T = ops.pack(m._wte, nE, d, 1, sk=d, sn=1, st=0, out=getattr(self, 'lm_T', None))       # dH = dlogits @ wte
            self._lm_T_stale = False
        else:
            self._lm_T_stale = True           # the step's own predicate also needs M1 % 64 == 0: its fallback must re-pack from THESE weights

    # ------------------------------------------------------------------ building blocks
    def _linear(self, x, name, M, res=None):
        return self.model._gemm(x, name, M, res=res)

    fuse_dropout = True               # bf16 arm: residual / MLP dropout in the projection GEMM's epilogue and in the LayerNorm backward's bf16 copy
                                      # (False: the separate dropout_add passes; the same masks, the same values up to the GEMM's own rounding)

    def _proj_dropout(self, x, name, M, res, drop):
        """res + dropout(x @ W + b)  (attn.c_proj -> resid_dropout, migt.py:216; mlp.c_proj -> the MLP's dropout, :72)"""
        dn = self.model._dense[name]
        if not drop[0]:
            return self._linear(x, name, M, res=res)
        if (self.fuse_dropout and dn.wp16 is not None and x.dtype == torch.bfloat16 and ops.gemm_drop_supported(M, dn.k, dn.n, drop[3])):
            out = torch.empty((M, dn.n), dtype=torch.float32, device=x.device)
            ops.igemm(x, dn.wp16, M, dn.k, dn.n, out, bias=dn.bias, res=res, bf16=True, a16=True, drop=drop)
            return out
        y = self._linear(x, name, M)
        return T.dropout_add(y, drop[0], drop[1], drop[2], res=res, out=y, row0=drop[3])

    def _linear_bwd(self, name, x, dy, M, need_dx=True, res=None, dx_bf16=False, gelu_bwd_u=None):
        """grads of y = x @ W + b given dy [M,N]; returns dx (+res) or None.  ``x`` may be a saved bf16 activation (bf16 arm);
        ``dx_bf16``: the bf16 arm's dX GEMM writes bf16 (the attention backward's dO operand)."""
        dn = self.model._dense[name]
        K, N = dn.k, dn.n
        bf16 = name in self.wpT16 and dn.wp16 is not None and M % 128 == 0
        first = self._first_write(name)                   # (lazy_gradient_zero: this layer's gradient pair holds last step's values, not zeros)
        if bf16 and self.tn_weight_gradient and ops.gemm_tn_bf16_supported(x, M, K, N):
            # bf16 arm with a saved bf16 activation: dW and db in ONE pass over x and dy as they lie (csrc/gemm_tn_bf16.hip) — no widening
            # transpose of x, no packed bf16 copy of dy, no column-sum pass
            gw, gb = self.g(name + '.weight'), self.g(name + '.bias')
            if self.overlap_weight_gradients and need_dx:
                # ... on a SECOND HIP stream, beside the dX GEMM that reads the same dy: both are one under-filled round of 256-tile
                # workgroups (225-243 on 256 CUs, one per CU: 128 KB of LDS each), so the weight gradient's workgroups take the CUs the
                # dX GEMM leaves idle and its tail.  Joined before anything reads the layer's gradients (_join_side)
                main = torch.cuda.current_stream(self.dev)
                side = self._side()
                side.wait_stream(main)                                                 # dy (and x) are complete
                with torch.cuda.stream(side):
                    ops.gemm_tn_bf16(x, dy, M, K, N, gw, gb, accumulate=not first, beside_another_gemm=True)
                for t in (x, dy):
                    t.record_stream(side)                                              # (the allocator must not recycle them under the side stream)
                self._side_busy = True
            else:
                # (alone on the compute stream the launch takes the full-machine split, another summation order; serial_wgrad_split_as_overlapped
                # keeps the second stream's split, for bit-for-bit comparisons of the two modes)
                ops.gemm_tn_bf16(x, dy, M, K, N, gw, gb, accumulate=not first, beside_another_gemm=self.serial_wgrad_split_as_overlapped and need_dx)
            return self._linear_dx(name, dy, M, res, dx_bf16, gelu_bwd_u) if need_dx else None
        if dy.dtype != torch.float32:
            raise RuntimeError('a bf16 gradient operand needs the TN weight-gradient path')
        if first:                                          # the paths below accumulate
            self.g(name + '.weight').zero_()
            self.g(name + '.bias').zero_()
        T.colsum(dy, self.g(name + '.bias'), M, N, accumulate=True)
        Mp = (M + 31) // 32 * 32                                                     # reduction length padded to the K stage
        xt = None
        if Mp != M:
            xt = torch.zeros((1, K, Mp), dtype=torch.float32, device=x.device)
        xt = T.transpose(x, M, K, out=xt, ld_dst=Mp)                                 # [K][Mp]
        gw = self.g(name + '.weight')
        x6 = name in self.wpT6 and dn.wp6 is not None and M % 64 == 0
        if bf16:
            # dW += X^T dY: (K/128)(N/128) = 36..144 output tiles for 256 CUs and a reduction of M = 19 200 rows -> split-K (as the x6
            # arm below); the split count must cut M into whole 64-row chunks of the packing
            tiles = ((K + 127) // 128) * ((N + 127) // 128)
            want = max(1, min(16, 768 // tiles))
            splits = max([s for s in range(1, want + 1) if M % (64 * s) == 0] or [1])
            if splits > 1 and (K * N) % 4 == 0:
                ops.gemm_bf16_splitk(xt, ops.pack_dense_kn_bf16(dy), K, M, N, gw, splits, lda=Mp)
            else:
                ops.igemm(xt, ops.pack_dense_kn_bf16(dy), K, M, N, gw, res=gw, lda=Mp, bf16=True)
        elif x6:
            # dW += X^T dY: [K][N] has only (K/128)(N/128) = 36..144 tiles for 256 CUs but a reduction of M = 19200 rows
            tiles = ((K + 127) // 128) * ((N + 127) // 128)
            splits = max(1, min(16, 768 // tiles, M // 1024))
            if splits > 1 and (K * N) % 4 == 0:
                ops.gemm_x6_splitk(xt, ops.pack_dense_kn_x6(dy), K, M, N, gw, splits, lda=Mp)
            else:
                ops.igemm(xt, ops.pack_dense_kn_x6(dy), K, M, N, gw, res=gw, lda=Mp, x6=True)
        else:
            dyp = ops.pack(dy, M, N, 1, sk=N, sn=1, st=0)                            # rows >= M are zero-filled by the packer
            ops.igemm(xt, dyp, K, Mp, N, gw, res=gw, lda=Mp)                         # dW += X^T dY
        if not need_dx:
            return None
        return self._linear_dx(name, dy, M, res, dx_bf16, gelu_bwd_u)

    serial_wgrad_split_as_overlapped = False
    overlap_weight_gradients = True   # bf16 arm: the TN weight-gradient GEMM of a layer on a second stream beside that layer's dX GEMM
    early_optimizer = False           # (measured in round 6: 19.660 vs 19.665 ms per step — the update hides, and the backward beside it slows by as much: off)
                                      # bf16 arm: a layer's AdamWeightDecay update and the re-packing of its weights are issued on a third stream as soon as
                                      # that layer's gradients are final (its backward and weight-gradient GEMMs done, its all-reduce complete) — HBM-bound
                                      # work beside the matrix-bound backward of the layers below instead of 0.65 ms at the end of the step.  Same kernels on
                                      # the same values: the parameters after a step are bit-identical (tests/test_train.py)
    fused_optimizer_repack = True     # bf16 arm: the optimizer step writes the dense layers' bf16 packings from the updated weights in the same pass
                                      # (vf_adamw_flat_pack_f32) instead of a re-pack launch that reads every weight again twice; bit-identical
    _adam_pack = None
    _packs_fresh = False
    _opt_stream_obj = None
    _layer_pack_ranges = None
    _early_layers_done = False
    _layer_nodecay = None
    _head_nodecay = None

    def _opt_stream(self):
        if self._opt_stream_obj is None:
            self._opt_stream_obj = torch.cuda.Stream(self.dev)
        return self._opt_stream_obj

    def _adam_scalars(self):
        c, step = self.cfg, self.step_count
        lr = learning_rate(step, self.lr_init, self.lr_total_steps, self.warmup_steps, self.lr_offset)
        t = step + 1
        return lr, lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)

    def _early_update_layer(self, i, handle=None):
        """AdamWeightDecay over layer i's range of the flat buffer + its eight bf16 packings, on the optimizer stream"""
        c = self.cfg
        if self._layer_nodecay is None:                                      # (built on the main stream, before the optimizer stream's wait below)
            self._layer_nodecay = []
            for a, b in self.layer_ranges:
                r = [[self.slices[n][0] - a, self.slices[n][1] - a] for n in self.names if 'bias' in n and a <= self.slices[n][0] < b]
                self._layer_nodecay.append(torch.tensor(sorted(r), dtype=torch.int64, device=self.dev).reshape(-1, 2))
        main, opt = torch.cuda.current_stream(self.dev), self._opt_stream()
        opt.wait_stream(main)                                                # the layer's backward (LayerNorm / bias gradients) is queued on main
        if self._side_busy:
            opt.wait_stream(self._side_stream)                               # ... and its weight-gradient GEMMs + slab sums on the side stream
        a, b = self.layer_ranges[i]
        lr, lr_adam = self._adam_scalars()
        with torch.cuda.stream(opt):
            if handle is not None:
                handle.wait()                                                # the layer's gradient all-reduce (this stream waits, not main)
            T.adamw_flat_(self.flat_p[a:b], self.flat_g[a:b], self.flat_m[a:b], self.flat_v[a:b],
                          self._layer_nodecay[i] if c.weight_decay > 0 else None, lr * c.weight_decay if c.weight_decay > 0 else 0.0, lr_adam,
                          self.b1, self.b2, self.eps)
            self._pack16(*self._layer_pack_ranges[i])

    overlap_attention_backward = False  # bf16 arm: the attention backward's dK / dV launch on a third stream beside its dQ launch — same bits, measured
                                        # 19.75 vs 19.65 ms per step (6 alternating rounds, round 6): the two kernels' workgroups sharing CUs cost more
                                        # than their tails; off
    _kv_stream_obj = None

    def _kv_stream(self):
        if self._kv_stream_obj is None:
            self._kv_stream_obj = torch.cuda.Stream(self.dev)
        return self._kv_stream_obj

    _side_stream = None
    _side_busy = False

    def _side(self):
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(self.dev)
        return self._side_stream

    def _join_side(self):
        """the compute stream waits for the weight-gradient stream (before a layer's gradients are reduced, clipped or applied)"""
        if self._side_busy:
            torch.cuda.current_stream(self.dev).wait_stream(self._side_stream)
            self._side_busy = False

    def _linear_dx(self, name, dy, M, res=None, dx_bf16=False, gelu_bwd_u=None):
        """dx = dy @ W^T (+ res).  ``gelu_bwd_u`` (bf16 arm): the saved fp32 pre-activation u of the GELU that produced this layer's input —
        the GEMM's epilogue then returns bf16(dx * gelu'(u)), the GELU backward without the fp32 dx ever reaching HBM"""
        dn = self.model._dense[name]
        K, N = dn.k, dn.n
        x6 = name in self.wpT6 and dn.wp6 is not None and M % 64 == 0
        bf16 = name in self.wpT16 and dn.wp16 is not None and M % 128 == 0
        if dx_bf16 and not (bf16 and res is None):
            raise RuntimeError('dx_bf16 needs the bf16 arm and no residual')
        dx = torch.empty((M, K), dtype=torch.bfloat16 if dx_bf16 else torch.float32, device=dy.device)
        if gelu_bwd_u is not None:
            if not (bf16 and dx_bf16 and res is None):
                raise RuntimeError('the fused GELU backward needs the bf16 arm, a bf16 result and no residual')
            u16 = gelu_bwd_u.dtype == torch.bfloat16
            ops.igemm(dy, self.wpT16[name], M, N, K, dx, res=gelu_bwd_u, epilogue=ops.EPI_GELU_BWD, bf16=True, a16=dy.dtype == torch.bfloat16, o16=True,
                      res16=u16, gelu_grad=u16 and self._u_is_derivative)       # (save_gelu_derivative: the tensor holds gelu'(u) already)
        elif bf16:
            ops.igemm(dy, self.wpT16[name], M, N, K, dx, res=res, bf16=True, a16=dy.dtype == torch.bfloat16, o16=dx_bf16)   # (bf16 dY: the 256-tile kernel)
        elif x6:
            ops.igemm(dy, self.wpT6[name], M, N, K, dx, res=res, x6=True)
        else:
            wpT = self.wpT.get(name)
            if wpT is None:                                                          # (a wide layer at a row count its arm does not tile)
                wpT = ops.pack(dn.w_raw, dn.n, dn.k, 1, sk=1, sn=dn.n, st=0)
            ops.igemm(dy, wpT, M, N, K, dx, res=res)
        return dx

    prune_last_block = True           # the last block's projection / LayerNorm / MLP (forward and backward) and ln_f on the branch streams' rows only: the main
                                      # stream's rows of the last block reach no loss (see train_step); its dropout masks there are indexed by the gathered rows
    lazy_gradient_zero = True         # the transformer layers' gradient tensors (85 M of the 88 M) are not zero-filled at the start of a step: each is written
                                      # exactly once per step, so its first writer STORES (sum_slabs / LayerNorm backward with accumulate off) instead of adding
                                      # to a zero — one 337 MB fill and one 337 MB read of zeros less per step.  Keys still unset when a layer's (or the
                                      # step's) backward ends are zero-filled then, so a tensor no kernel wrote is a zero gradient, as before.  The head range
                                      # (embeddings, pose heads, ln_f: several contributions each) is zero-filled as before.  Same bits: x + 0 == x.
    _unset = frozenset()
    _layer_grad_keys = None

    def _begin_gradients(self):
        if not (self.lazy_gradient_zero and self.layer_ranges):
            self.flat_g.zero_()
            self._unset = set()
            return
        if self._layer_grad_keys is None:
            lo = self.layer_ranges[0][0]
            self._layer_grad_keys = ([n[:-7] for n in self.names if n.endswith('.weight') and self.slices[n][0] >= lo and n[:-7] + '.bias' in self.slices]
                                     + [n[:-6] for n in self.names if n.endswith('.gamma') and self.slices[n][0] >= lo and n[:-6] + '.beta' in self.slices])
            covered = sum(self.slices[k + a][1] - self.slices[k + a][0] for k in self._layer_grad_keys
                          for a in (('.weight', '.bias') if k + '.weight' in self.slices else ('.gamma', '.beta')))
            if covered != sum(self.slices[n][1] - self.slices[n][0] for n in self.names if self.slices[n][0] >= lo):
                raise RuntimeError('lazy_gradient_zero: a layer tensor is neither a dense layer\'s weight | bias nor a LayerNorm\'s gamma | beta')
        a, b = self.head_range
        self.flat_g[a:b].zero_()
        self._unset = set(self._layer_grad_keys)

    def _first_write(self, key):
        """True once per step for a layer tensor pair that was not zero-filled: its writer must store, not accumulate"""
        if key in self._unset:
            self._unset.discard(key)
            return True
        return False

    def _flush_unset(self, prefix=''):
        """zero-fill what no kernel wrote (keys under ``prefix``)"""
        for k in [k for k in self._unset if k.startswith(prefix)]:
            for a in (('.weight', '.bias') if k + '.weight' in self.slices else ('.gamma', '.beta')):
                self.g(k + a).zero_()
            self._unset.discard(k)

    def _ln_bwd(self, name, dy, x, M, res=None, also_bf16=False, drop=(0.0, 0, 0)):
        d = self.cfg.d_model
        return T.layernorm_bwd(dy, x, self.p(name + '.gamma'), self.g(name + '.gamma'), self.g(name + '.beta'), M, d, accumulate=not self._first_write(name),
                               res=res, also_bf16=also_bf16, drop=drop if also_bf16 else (0.0, 0, 0))

    bf16_preactivation = True         # bf16 arm, with both GELU fusions: c_fc's pre-activation u is SAVED as bf16 (the reference's mixed_float16 policy
                                      # keeps every activation in half precision); gelu(u) is still taken from the fp32 accumulator, gelu'(u)
                                      # in the backward epilogue from the rounded u.  False: u saved as fp32.
    save_gelu_derivative = True       # bf16 arm, with bf16_preactivation: what c_fc saves for the backward is gelu'(u) (bf16, from the erf / exp evaluation its GELU
                                      # makes anyway: three more instructions per element) instead of u, whose only reader — the GELU-backward epilogue of the
                                      # mlp.c_proj dX GEMM — then multiplies by a loaded value instead of evaluating erf + exp per element again (that epilogue
                                      # was 22 us of a 148 us launch).  gelu' is taken from the fp32 pre-activation and rounded once; before it was evaluated
                                      # on the bf16-rounded u
    _u_is_derivative = False
    fuse_gelu_forward = True          # bf16 arm: c_fc writes u (fp32, saved) and bf16 gelu(u) from one epilogue (VF_EPI_GELU_DUAL); the GELU there
                                      # is the inference arm's vf_gelu_erf_fast (|err| 1.5e-7: a few outputs round to the neighbouring bf16)

    def _gelu_dual_ok(self, M):
        c = self.cfg
        return M >= 256 and c.d_model % 128 == 0 and (4 * c.d_model) % 256 == 0 and M * 4 * c.d_model * 2 < 2 ** 31   # (the 256-tile kernel's limits)

    bf16_residual_gradient = True     # bf16 arm: the LayerNorm backward also writes its result as bf16 — the operand of the two projection
                                      # layers' backward GEMMs, which then run on the 256-tile kernel (the rounding is the one the GEMM's operand
                                      # load applied anyway; only the two bias gradients see it)

    small_n_pose_head = True          # the pose head's 1536 -> 7 layer and its dW on the one-pass small-N kernels (False: the implicit-GEMM kernel, round 5)
    bf16_lm_head = True               # bf16 arm: the tied LM head (logits, dH, dwte) on the bf16 pipe like every other wide layer (the native-f32
                                      # GEMMs it replaces were 0.8 ms of a 21 ms step); False: native f32 MFMA
    _lm16 = None
    _lm_T_stale = True
    one_launch_repack = True          # bf16 arm: all weight packings refreshed by one launch per step
    _pack16 = None
    _pack16_keep = None
    fuse_gelu_backward = True         # bf16 arm: d(pre-activation) = dX(mlp.c_proj) * gelu'(u) in that GEMM's epilogue (same bits as the two passes)
    bf16_gradient_operands = True     # bf16 arm: gelu_bwd / attention backward write their gradients as bf16 (see train_step)
    tn_weight_gradient = True         # bf16 arm: dW / db of the wide layers straight from the row-major operands (False: transpose + pack + column sums)
    bf16_saved_activations = True     # bf16 arm: LayerNorm outputs / MLP hidden saved as bf16 (see train_step); False keeps them fp32 (same gradients)
    attention_arith = 'bf16'          # bf16 arm only: 'bf16' = attention forward / backward on the bf16 matrix pipe; 'f32' = the exact-f32 kernels
    fused_optimizer = True            # AdamWeightDecay as ONE launch over the flat buffer (False: one launch per tensor, 468 per step; bit-identical)
    _nodecay = None
    attention_backward = 'flash'      # 'dense': the first version (P materialised per head with batched GEMMs), kept for A/B

    scene_offset = None               # index of this rank's first scene in the GLOBAL batch of a data-parallel step (None: rank * local batch).
                                      # Every random draw of the step — the dropout masks, the random pose multiplier — is a function of
                                      # the GLOBAL scene index, so N ranks draw exactly what one process draws on the concatenated batch,
                                      # and no two replicas share a mask (with one seed per step and local indices they all would)

    def _scene_offset(self, B):
        if self.scene_offset is not None:
            return int(self.scene_offset)
        return (dist.get_rank(self.group) * B) if self._world() > 1 else 0

    def random_pose_factors(self, B, seed, b0=0):
        """rpm ** u_b, u_b = 2 * hash(seed, SITE_POSE_MULT, b0 + b) / 2^32 - 1 (migt.py:351; counter-based like the dropout masks)"""
        from ._hash import dropout_hash
        u = dropout_hash(seed, SITE_POSE_MULT, np.arange(b0, b0 + B, dtype=np.uint64)).astype(np.float64) / 2.0 ** 32 * 2.0 - 1.0
        return torch.from_numpy((float(self.cfg.random_pose_multiplier) ** u).astype(np.float32))

    def step_seed(self, step):
        """per-step dropout seed (uint32) from ``dropout_seed`` and the step counter"""
        return (int(self.dropout_seed) * 0x9E3779B1 + int(step) * 0x85EBCA77 + 0x1234567) & 0xFFFFFFFF

    def _attn_bwd(self, qkv, datt, B, Tn, L, spec, att=None, lse=None, drop=(0.0, 0, 0)):
        """dQKV from dA (branching_attention.py:82-126 semantics).  'flash': one pair of kernels per layer re-materialises the
        probabilities tile by tile from the saved log-sum-exp and skips masked tiles; 'dense': per head with batched GEMMs."""
        c = self.cfg
        d, H = c.d_model, c.n_head
        M = B * Tn
        dqkv = torch.empty((M, 3 * d), dtype=torch.float32, device=qkv.device)
        if drop[0] and not (self.attention_backward == 'flash' and att is not None and lse is not None):
            raise NotImplementedError("attention dropout needs attention_backward='flash'")
        if self.attention_backward == 'flash' and att is not None and lse is not None:
            # thirds are (V, Q, K) (migt.py:207-213): gradients land in the same thirds of dqkv
            T.attn_bwd(qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d], att, datt, lse,
                       dqkv[:, d:2 * d], dqkv[:, 2 * d:], dqkv[:, :d], B, H, Tn, L,
                       3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, spec, drop=drop)
            return dqkv
        S = torch.empty((B, Tn, Tn), dtype=torch.float32, device=qkv.device)
        dP = torch.empty_like(S)
        pf_T = ops.packed_floats(64, Tn)
        bs = Tn * 3 * d
        for h in range(H):
            q, k, v = qkv[:, d + h * 64:], qkv[:, 2 * d + h * 64:], qkv[:, h * 64:]
            da = datt[:, h * 64:]
            kp = ops.pack(k, 64, Tn, 1, sk=1, sn=3 * d, st=0, batch=B, src_bstride=bs)              # B[c][key] = K[key][c]
            ops.igemm(q, kp, Tn, 64, Tn, S, lda=3 * d, batch=B, stride_x=bs, stride_w=pf_T, stride_out=Tn * Tn)
            T.softmax_mask_(S, B, Tn, L, spec, 1.0)                                                   # P
            vp = ops.pack(v, 64, Tn, 1, sk=1, sn=3 * d, st=0, batch=B, src_bstride=bs)              # B[c][key] = V[key][c]
            ops.igemm(da, vp, Tn, 64, Tn, dP, lda=d, batch=B, stride_x=Tn * d, stride_w=pf_T, stride_out=Tn * Tn)
            T.softmax_mask_bwd_(S, dP, B, Tn, L, spec, 1.0)                                           # dS (in dP)
            pf_k = ops.packed_floats(Tn, 64)
            kq = ops.pack(k, Tn, 64, 1, sk=3 * d, sn=1, st=0, batch=B, src_bstride=bs)              # B[key][c] = K[key][c]
            ops.igemm(dP, kq, Tn, Tn, 64, dqkv[:, d + h * 64:], lda=Tn, ldc=3 * d, batch=B, stride_x=Tn * Tn,
                      stride_w=pf_k, stride_out=bs)                                                   # dQ = dS K
            dSt = T.transpose(dP, Tn, Tn, batch=B, bs_src=Tn * Tn)
            qq = ops.pack(q, Tn, 64, 1, sk=3 * d, sn=1, st=0, batch=B, src_bstride=bs)
            ops.igemm(dSt, qq, Tn, Tn, 64, dqkv[:, 2 * d + h * 64:], lda=Tn, ldc=3 * d, batch=B, stride_x=Tn * Tn,
                      stride_w=pf_k, stride_out=bs)                                                   # dK = dS^T Q
            Pt = T.transpose(S, Tn, Tn, batch=B, bs_src=Tn * Tn)
            dap = ops.pack(da, Tn, 64, 1, sk=d, sn=1, st=0, batch=B, src_bstride=Tn * d)
            ops.igemm(Pt, dap, Tn, Tn, 64, dqkv[:, h * 64:], lda=Tn, ldc=3 * d, batch=B, stride_x=Tn * Tn,
                      stride_w=pf_k, stride_out=bs)                                                   # dV = P^T dA
        return dqkv

    # ------------------------------------------------------------------ the step
    def train_step(self, poses, tokens, reduce_gradients: bool = True, apply_update: bool = True, _forward_only: bool = False):
        """poses [b,S,7] float32 (already through process_batch: relative + normalised), tokens [b,S,t,t] int.
        Returns the metrics dict of MIGT.train_step (loss, ce_loss, acc, pose_* ...).  ``_forward_only`` (test_step / predict_step):
        the same multi-stream graph with ``training=False`` — no dropout, no random pose multiplier — up to the losses; returns
        ``(metrics, outputs)``."""
        m, c, dev = self.model, self.cfg, self.dev
        poses = poses.to(dev)
        tokens = tokens.to(dev)
        if poses.dtype != torch.float32:
            raise TypeError('poses must be float32 (migt.py:346)')
        B, S = tokens.shape[:2]
        L = int(np.prod(tokens.shape[2:]))
        d, H, nE = c.d_model, c.n_head, c.n_embeddings
        use_loc = m.use_localization
        NS = 3 if use_loc else 2
        V = NS * S
        Tn, M, M1 = V * L, B * V * L, B * S * L
        skip = c.n_loss_skip
        if not 0 <= skip < S:
            raise ValueError('n_loss_skip must be < sequence length')
        if S < 2:
            raise ValueError('train_step needs sequences of at least 2 views (the STREAMS attention mask encodes the stream '
                             'length as -S <= -2; a 1-view sequence has nothing to condition on)')
        if not _forward_only:
            self._begin_gradients()

        # ---- forward with saved activations --------------------------------------------------------------
        seed = self.step_seed(self.step_count)
        rmul = None
        pin = geometry.pose_model_input(poses, c.pose_multiplier)
        if c.random_pose_multiplier != 1 and not _forward_only:                      # migt.py:350-354: rpm ** U(-1, 1) per scene (training only)
            rmul = self.random_pose_factors(B, seed, self._scene_offset(B)).to(dev)
            pin = torch.cat([pin[..., :3] * rmul.view(B, 1, 1), pin[..., 3:]], -1)
        pin = pin.reshape(B * S, 7).contiguous()
        fc = m._dense['pose_embedding.c_fc']
        u1 = ops.dense_small_k(pin, fc.w_raw, fc.bias, B * S, 7, fc.n, gelu=False)
        h1 = T.gelu(u1)
        pe = self._linear(h1, 'pose_embedding.c_proj', B * S).view(B, S, d)
        tok = tokens.reshape(B, S, L)
        ids_streams = [tok, torch.full_like(tok, m.mask_token)]
        add_streams = [pe, pe]
        if use_loc:
            ids_streams.append(tok)
            add_streams.append(m._wte[m.localization_token].view(1, 1, d).expand(B, S, d))
        ids32 = torch.cat(ids_streams, 1).reshape(M).to(torch.int32).contiguous()
        add = torch.cat(add_streams, 1).contiguous().view(B * V, d)
        h = ops.embed_sum(ids32, m._wte, m._wpe, add, B * V, L, d, nE + 2)
        rate = 0.0 if _forward_only else float(c.dropout)                            # Dropout layers are inert with training=False
        b0 = self._scene_offset(B) if rate else 0
        row0, plane0 = b0 * V * L, b0 * H                                            # this rank's first row / first attention plane in the global batch
        if rate:
            T.dropout_add(h, rate, seed, SITE_EMBED, out=h, row0=row0)               # self.drop, migt.py:403
        saved = []
        # attention of the bf16 arm on the bf16 matrix pipe (csrc/attention_dma.hip + attention_train_bf16.hip) where its kernels apply:
        # 64-token views, the wide c_attn / c_proj layers on their bf16 packings; otherwise the exact-f32 kernels.  Attention dropout is
        # inside both sets of kernels (the same masks)
        ca, cp = m._dense['h.0.attn.c_attn'], m._dense['h.0.attn.c_proj']
        attn16 = (self.attention_arith == 'bf16' and m.precision == 'bf16' and T.attn_bf16_supported(Tn, L)
                  and d // H == 64 and ca.wp16 is not None and cp.wp16 is not None and M % 128 == 0
                  and (rate == 0.0 or Tn * (Tn // 4) < 2 ** 32))
        # bf16 arm, wide layers: the activations only GEMMs read — both LayerNorm outputs, the attention output, the MLP hidden — are
        # SAVED AS bf16 by their producers (the rounding the GEMM applied on load before: identical products), so the forward GEMMs take
        # the 256-tile LDS-DMA kernel (bf16 A operand) and the saved activations halve; the backward reads them through the widening
        # transpose (dW) and never otherwise (LayerNorm / GELU backward use the fp32 h / u)
        # ... and the two gradients only GEMMs read — d(MLP pre-activation) from the GELU backward, d(q | k | v) from the attention backward —
        # are WRITTEN as bf16 by those kernels (again the rounding their consumers applied on load: identical dX / dW; the bias gradients
        # become sums of the rounded values), so that dX takes the 256-tile kernel and the TN kernel moves half the bytes
        act16 = attn16 and self.bf16_saved_activations and all(m._dense[f'h.0.{n}'].wp16 is not None for n in ('mlp.c_fc', 'mlp.c_proj'))
        # bf16 gradient operands need every consumer that can read one: the TN weight-gradient kernel for all four layer shapes of a block
        # at this M (K, N multiples of 256, 32-bit offsets) and the 256-tile GEMM for the dX products (an explicit predicate: d_model 384
        # or 640 passes the packing rule of `act16` but not these, and keeps fp32 gradients + the transpose / pack weight-gradient path)
        block_shapes = [(m._dense[f'h.0.{n}'].k, m._dense[f'h.0.{n}'].n) for n in ('attn.c_attn', 'attn.c_proj', 'mlp.c_fc', 'mlp.c_proj')]
        grad16 = (act16 and self.bf16_gradient_operands and self.tn_weight_gradient
                  and all(ops.gemm_tn_bf16_shape_ok(M, k_, n_) and ops.gemm_g256_shape_ok(M, n_, k_) for k_, n_ in block_shapes))
        # the LayerNorm backward hands the projection layers' backward GEMMs a bf16 copy of the residual-stream gradient (with dropout: under
        # the consuming layer's output mask, which only the fused form applies there)
        # (with dropout the masks ride in layernorm_bwd(drop=...), which indexes 32-bit mask groups: the same bound gemm_drop_supported puts on the
        # forward — a global batch so large that (M + row0) / 4 * d passes 2^32 falls back to the dropout_add passes instead of raising)
        res16 = (grad16 and self.bf16_residual_gradient and self.fuse_gelu_backward
                 and (rate == 0.0 or (self.fuse_dropout and row0 % 4 == 0 and ((M + row0 + 3) // 4) * d < 2 ** 32)))
        drop_of = lambda site: (rate, seed, site, row0)                              # noqa: E731  (elementwise sites: row offset)
        drop_attn = lambda i_: (rate, seed, site_attn(i_), plane0)                   # noqa: E731  (attention: plane offset)
        # prune_last_block: the losses read the branch streams only (MASK -> LM head, LOC -> pose head: migt.py:416-448); every block but the last needs
        # the main stream's rows as keys and values of the next one, the LAST block's projection / LayerNorm / MLP on them feed nothing — and their
        # backward multiplies zeros.  From the last block's attention on, forward and backward run on the NS - 1 branch streams' rows (gathered:
        # they are contiguous per scene), ln_f with them; the attention backward and everything below it get the two gradients scattered back with
        # zeros in the main stream's rows, which is what they were.  The dropout masks of the last block's two sites are indexed by the gathered rows
        nb = NS - 1
        tail = bool(self.prune_last_block and NS > 1 and c.n_layer > 0)
        Mx = B * nb * S * L if tail else M                                           # rows from the last block's projection on
        drop_x = (lambda site: (rate, seed, site, row0 // NS * nb)) if tail else drop_of      # noqa: E731
        for i in range(c.n_layer):
            p = f'h.{i}'
            n1 = ops.layernorm(h, *m._ln[p + '.ln_1'], M, d, out_bf16=act16)
            if attn16:
                # bf16 arm: c_attn writes q | k | v as bf16, the LDS-DMA kernel of the inference arm (+ log-sum-exp) writes a bf16 output
                # that c_proj reads as it is (the rounding the GEMM would apply on load)
                qkv = m._gemm(n1, p + '.attn.c_attn', M, out_bf16=True)
                att = torch.empty((M, d), dtype=torch.bfloat16, device=dev)
                lse = T.attn_fwd_lse_bf16(qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d], att, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, -S,
                                          drop=drop_attn(i))
            else:
                qkv = self._linear(n1, p + '.attn.c_attn', M)
                att = torch.empty((M, d), dtype=torch.float32, device=dev)
                lse = T.attn_fwd_lse(qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d], att, B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, 1.0, -S,
                                     drop=drop_attn(i))
            last = tail and i == c.n_layer - 1
            Mi, drop_i = (Mx, drop_x) if last else (M, drop_of)
            att_i, h_i = att, h
            if last:
                att_i = att.view(B, NS, S * L, att.shape[-1])[:, 1:].reshape(Mx, att.shape[-1])
                h_i = h.view(B, NS, S * L, d)[:, 1:].reshape(Mx, d)
            h_mid = self._proj_dropout(att_i, p + '.attn.c_proj', Mi, h_i, drop_i(site_resid(i)))      # h + resid_dropout(c_proj(a)), migt.py:216,233
            n2 = ops.layernorm(h_mid, *m._ln[p + '.ln_2'], Mi, d, out_bf16=act16)
            if act16 and self.fuse_gelu_forward and self._gelu_dual_ok(Mi):
                # c_fc keeps the fp32 pre-activation for the backward pass AND hands bf16 gelu(u) to mlp.c_proj from one epilogue
                dn = m._dense[p + '.mlp.c_fc']
                u16 = res16 and self.bf16_preactivation      # (its only reader then: the
                # GELU-backward epilogue of the 256-tile kernel, fed by the bf16 residual-stream gradient)
                u = torch.empty((Mi, dn.n), dtype=torch.bfloat16 if u16 else torch.float32, device=dev)
                f = torch.empty((Mi, dn.n), dtype=torch.bfloat16, device=dev)
                u_deriv = bool(u16 and self.save_gelu_derivative)                         # (`u` then holds gelu'(u): see save_gelu_derivative)
                ops.igemm(n2, dn.wp16, Mi, dn.k, dn.n, u, bias=dn.bias, epilogue=ops.EPI_GELU_DUAL, bf16=True, a16=True, o16=u16, out_aux=f,
                          gelu_grad=u_deriv)
            else:
                u_deriv = False
                u = self._linear(n2, p + '.mlp.c_fc', Mi)
                f = T.gelu(u, out_bf16=act16)
            h_out = self._proj_dropout(f, p + '.mlp.c_proj', Mi, h_mid, drop_i(site_mlp(i)))           # h + dropout(mlp(...)), migt.py:72,237
            if not _forward_only:
                saved.append((h, n1, qkv, att, att_i, h_mid, n2, u, f, lse, u_deriv))
            h = h_out
        so = 1 if tail else 0                                                         # (hf's first stream: the main stream is not there when pruned)
        hf = ops.layernorm(h, *m._ln['ln_f'], Mx, d).view(B, NS - so, S, L, d)

        # ---- losses (migt.py:416-448) ---------------------------------------------------------------------
        view_ok = (torch.arange(S, device=dev) >= skip).float().view(1, S, 1).expand(B, S, L).reshape(M1)
        denom = float((S - skip) * L * B)
        hmask = hf[:, 1 - so].contiguous().view(M1, d)
        logits = torch.empty((M1, nE), dtype=torch.float32, device=dev)
        lm16 = self._lm16 is not None and m.precision == 'bf16' and self.bf16_lm_head and M1 % 64 == 0
        if lm16:
            ops.igemm(hmask, self._lm16[0], M1, d, nE, logits, bf16=True)
        else:
            ops.igemm(hmask, m._lm_head, M1, d, nE, logits)
        tgt = tok.reshape(M1).to(torch.int32).contiguous()
        w_ce = (view_ok * (c.image_generation_weight / denom)).contiguous()
        ce_rows, dlogits = T.softmax_ce(logits, tgt, w_ce, M1, nE, c.label_smoothing)        # :420-423
        ce_b = ce_rows.view(B, S, L)[:, skip:].mean((1, 2))
        loss_b = ce_b * c.image_generation_weight
        metrics = dict(ce_loss=ce_b.mean())
        pred = ops.argmax_rows(logits, M1, nE).view(B, S, L)
        metrics['acc'] = (pred[:, skip:] == tok[:, skip:]).float().mean()
        if use_loc:
            loc_w = float(self.loc_weight(self.step_count))
            hloc = hf[:, 2 - so].contiguous().view(M1, d)
            up = self._linear(hloc, 'pose_criterion.pose_classifier.c_fc', M1)
            p1 = T.gelu(up)
            dn_p = m._dense['pose_criterion.pose_classifier.c_proj']
            small_n = self.small_n_pose_head and T.dense_small_n_supported(dn_p.k, dn_p.n) and p1.dtype == torch.float32 and p1.is_contiguous()
            if small_n:          # 1536 -> 7: one pass over p1 instead of an implicit GEMM on a 32-column tile (176 -> ~15 us, round 6)
                raw = T.dense_small_n(p1, dn_p.w_raw, dn_p.bias, M1, dn_p.k, dn_p.n)
            else:
                raw = self._linear(p1, 'pose_criterion.pose_classifier.c_proj', M1)
            dyn = c.use_dynamic_pose_loss
            if dyn:                                                                  # DynamicLossWeightingCriterion, migt.py:107-120
                sw = self.p(DYN_KEY)
                e_pos, e_ori = float(torch.exp(-sw[0])), float(torch.exp(-sw[1]))
                rows_per_scene = float((S - skip) * L)
                w_pos = (view_ok * (loc_w * e_pos / rows_per_scene)).contiguous()    # reduce_SUM over the batch (:117): no 1/B
                w_ori = (view_ok * (loc_w * e_ori / rows_per_scene)).contiguous()
            else:
                w_pos = w_ori = (view_ok * (loc_w / denom)).contiguous()
            div = None if rmul is None else rmul.view(B, 1).expand(B, S * L).reshape(M1).contiguous()
            pos_r, ori_r, draw = T.pose_mse(raw, poses.reshape(B * S, 7).contiguous(), w_pos, M1, L, c.pose_multiplier, w_ori=w_ori,
                                            xyz_div=div)
            pos_b = pos_r.view(B, S, L)[:, skip:].mean((1, 2))
            ori_b = ori_r.view(B, S, L)[:, skip:].mean((1, 2))
            if dyn:
                pose_loss = (sw[0] + e_pos * pos_b + sw[1] + e_ori * ori_b).sum()    # a scalar, broadcast onto every scene (:440,447)
                loss_b = loss_b + pose_loss * loc_w
                gs = self.g(DYN_KEY)
                gs[0] = loc_w * (B - e_pos * pos_b.sum())
                gs[1] = loc_w * (B - e_ori * ori_b.sum())
                metrics.update(dynamic_loss_weight_pos=sw[0].clone(), dynamic_loss_weight_ori=sw[1].clone(), pose_loss=pose_loss)
            else:
                loss_b = loss_b + (pos_b + ori_b) * loc_w
                metrics['pose_loss'] = (pos_b + ori_b).mean()
            metrics.update(pose_pos_loss=pos_b.mean(), pose_ori_loss=ori_b.mean(), localization_weight=loc_w)
        metrics['loss'] = loss_b.mean()                                              # reduce_mean, migt.py:476
        if _forward_only:
            return metrics, dict(logits=logits.view(B, S, L, nE), predicted_tokens=pred,
                                 pose_head_raw=raw.view(B, S, L, 7) if use_loc else None)

        # ---- backward -------------------------------------------------------------------------------------
        dhf = torch.zeros((B, NS - so, S, L, d), dtype=torch.float32, device=dev)
        # tied LM head: dH = dlogits @ wte[:nE];  dwte[:nE] = dlogits^T @ H
        dhm = torch.empty((M1, d), dtype=torch.float32, device=dev)
        gwte = self.g('wte.weight')
        if lm16:
            ops.igemm(dlogits, self._lm16[1], M1, nE, d, dhm, bf16=True)
            # dwte[:nE] = dlogits^T @ H straight from the row-major operands (the TN kernel: x = dlogits as bf16, dy = H)
            ops.gemm_tn_bf16(dlogits.to(torch.bfloat16), hmask, M1, nE, d, gwte[:nE], None)
        else:
            if getattr(self, 'lm_T', None) is None or self._lm_T_stale:               # (M1 % 64 != 0 in the bf16 arm: repack() skipped the native
                self.lm_T = ops.pack(m._wte, nE, d, 1, sk=d, sn=1, st=0,              # transposed packing — build it from the CURRENT weights, every
                                     out=getattr(self, 'lm_T', None))                 # step: a packing cached from step 1 would be silently stale)
                self._lm_T_stale = False
            ops.igemm(dlogits, self.lm_T, M1, nE, d, dhm)
            dlt = T.transpose(dlogits, M1, nE)
            hp = ops.pack(hmask, M1, d, 1, sk=d, sn=1, st=0)
            ops.igemm(dlt, hp, nE, M1, d, gwte)                                       # rows [0, nE) of the (zeroed) grad
        dhf[:, 1 - so] = dhm.view(B, S, L, d)
        if use_loc:
            name = 'pose_criterion.pose_classifier.c_proj'
            dn = m._dense[name]
            T.colsum(draw, self.g(name + '.bias'), M1, 7, accumulate=True)
            if small_n:          # dW = p1^T @ draw straight from the row-major operands (no transpose, no packing)
                T.dense_small_n_wgrad(p1, draw, self.g(name + '.weight'), M1, dn.k, 7, accumulate=True)
            else:
                p1t = T.transpose(p1, M1, dn.k)
                drp = ops.pack(draw, M1, 7, 1, sk=7, sn=1, st=0)
                ops.igemm(p1t, drp, dn.k, M1, 7, self.g(name + '.weight'))
            w2t = T.transpose(dn.w_raw, dn.k, 7).view(7, dn.k)
            dp1 = ops.dense_small_k(draw, w2t, None, M1, 7, dn.k, gelu=False)
            dup = T.gelu_bwd(up, dp1)
            dhl = self._linear_bwd('pose_criterion.pose_classifier.c_fc', hloc, dup, M1)
            dhf[:, 2 - so] = dhl.view(B, S, L, d)
        # with dropout, the bf16 copy of a residual-stream gradient is the dY of the projection layer that consumes it, i.e. that gradient
        # under the layer's OUTPUT mask: the LayerNorm backward applies it while it writes the copy (the fp32 gradient stays unmasked)
        nl = c.n_layer
        dh = self._ln_bwd('ln_f', dhf.view(Mx, d), h, Mx, also_bf16=res16 and nl > 0, drop=drop_x(site_mlp(nl - 1)))
        dh, dh16 = dh if (res16 and nl > 0) else (dh, None)
        handles = []
        overlap = reduce_gradients and self._world() > 1 and not (c.gradient_clip_val and c.gradient_clip_val > 0)
        # early per-layer optimizer (see early_optimizer): needs the one-launch optimizer and re-pack tables, no clipping pass over the finished
        # gradients, and — with more than one rank — the per-layer fp32 all-reduce whose handle the optimizer stream can wait for
        early = (self.early_optimizer and apply_update and self.fused_optimizer and self._pack16 is not None and self._layer_pack_ranges is not None
                 and not (c.gradient_clip_val and c.gradient_clip_val > 0)
                 and (self._world() == 1 or not reduce_gradients or (overlap and self.grad_allreduce_dtype == 'f32')))
        for i in reversed(range(c.n_layer)):
            p = f'h.{i}'
            h_in, n1, qkv, att, att_i, h_mid, n2, u, f, lse, self._u_is_derivative = saved[i]
            last = tail and i == c.n_layer - 1
            Mi, drop_i = (Mx, drop_x) if last else (M, drop_of)
            dy_mlp = dh16 if res16 else (T.dropout_add(dh, rate, seed, site_mlp(i), row0=drop_i(0)[3]) if rate else dh)      # d(mlp.c_proj output)
            if grad16 and self.fuse_gelu_backward:                                   # GELU backward in the epilogue of the dX GEMM that feeds it
                du = self._linear_bwd(p + '.mlp.c_proj', f, dy_mlp, Mi, dx_bf16=True, gelu_bwd_u=u)
            else:
                df = self._linear_bwd(p + '.mlp.c_proj', f, dy_mlp, Mi)
                du = T.gelu_bwd(u, df, out_bf16=grad16)
            dn2 = self._linear_bwd(p + '.mlp.c_fc', n2, du, Mi)
            dh_mid = self._ln_bwd(p + '.ln_2', dn2, h_mid, Mi, res=dh, also_bf16=res16, drop=drop_i(site_resid(i)))  # (+ the residual branch's gradient)
            dh_mid, dh_mid16 = dh_mid if res16 else (dh_mid, None)
            datt = self._linear_bwd(p + '.attn.c_proj', att_i,
                                    dh_mid16 if res16 else (T.dropout_add(dh_mid, rate, seed, site_resid(i), row0=drop_i(0)[3]) if rate else dh_mid), Mi,
                                    dx_bf16=attn16)
            if last:
                # back to all rows for the attention backward and the block's first half: the main stream's rows of d(attention output) and of the
                # residual-stream gradient are zeros (nothing downstream of them reached a loss)
                full = torch.zeros((B, NS, S * L, datt.shape[-1]), dtype=datt.dtype, device=dev)
                full[:, 1:] = datt.view(B, nb, S * L, datt.shape[-1])
                datt = full.view(M, datt.shape[-1])
                full = torch.zeros((B, NS, S * L, d), dtype=torch.float32, device=dev)
                full[:, 1:] = dh_mid.view(B, nb, S * L, d)
                dh_mid = full.view(M, d)
            if attn16:
                dqkv = torch.empty((M, 3 * d), dtype=torch.bfloat16 if grad16 else torch.float32, device=dev)
                T.attn_bwd_bf16(qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d], att, datt, lse, dqkv[:, d:2 * d], dqkv[:, 2 * d:], dqkv[:, :d],
                                B, H, Tn, L, 3 * d, 3 * d, 3 * d, d, d, 3 * d, 3 * d, 3 * d, 1.0, -S, drop=drop_attn(i),
                                kv_stream=self._kv_stream() if self.overlap_attention_backward else None)
            else:
                dqkv = self._attn_bwd(qkv, datt, B, Tn, L, -S, att=att, lse=lse, drop=drop_attn(i))
            dn1 = self._linear_bwd(p + '.attn.c_attn', n1, dqkv, M)
            dh = self._ln_bwd(p + '.ln_1', dn1, h_in, M, res=dh_mid, also_bf16=res16 and i > 0, drop=drop_of(site_mlp(i - 1)))
            dh, dh16 = dh if (res16 and i > 0) else (dh, None)
            saved[i] = None
            hd = None
            self._flush_unset(p + '.')
            if overlap:                                                              # this layer's grads are final
                self._join_side()
                hd = self._allreduce_range(*self.layer_ranges[i])
                handles.append(hd)
            if early:
                self._early_update_layer(i, hd[0] if hd is not None else None)
        # embeddings: dwte scatter, dwpe, d(add) -> pose embedding MLP / LOC token row
        if rate:
            T.dropout_add(dh, rate, seed, SITE_EMBED, out=dh, row0=row0)
        dadd = T.embed_bwd(dh, ids32, gwte, self.g('wpe.embeddings'), B * V, L, d, nE + 2).view(B, V, d)
        dpe = (dadd[:, :S] + dadd[:, S:2 * S]).contiguous().view(B * S, d)
        if use_loc:
            T.colsum(dadd[:, 2 * S:].contiguous().view(B * S, d), gwte[m.localization_token], B * S, d, accumulate=True)
        dh1 = self._linear_bwd('pose_embedding.c_proj', h1, dpe, B * S)
        du1 = T.gelu_bwd(u1, dh1)
        T.dense_small_k_bwd(pin, du1, self.g('pose_embedding.c_fc.weight'), self.g('pose_embedding.c_fc.bias'), B * S, 7, fc.n)

        # ---- clip (per replica, per tensor, before aggregation), all-reduce SUM, AdamWeightDecay ------------
        self._flush_unset()
        self._join_side()
        if c.gradient_clip_val and c.gradient_clip_val > 0:
            for n in self.names:
                T.clip_by_norm_(self.g(n).reshape(-1), float(c.gradient_clip_val), self._scratch)
        if reduce_gradients and self._world() > 1:
            if overlap:
                handles.append(self._allreduce_range(*self.head_range))
            elif self.grad_allreduce_dtype == 'bf16':
                handles += [self._allreduce_range(a, b) for a, b in [self.head_range] + self.layer_ranges]
            else:
                handles += [(h, None, None) for h in
                            sharding.allreduce_sum_ranges(self.flat_g, [self.head_range] + self.layer_ranges, self.group)]
            ev = None
            if self.time_allreduce:                                                  # what the backward pass did NOT hide: the compute
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))   # stream's wait for the collectives
                ev[0].record()
            for hd, buf, rng in handles:
                hd.wait()
                if buf is not None:                                                  # bf16 bucket: back into the fp32 gradient buffer
                    self.flat_g[rng[0]:rng[1]].copy_(buf)
            if ev is not None:
                ev[1].record()
                self._allreduce_events = ev
        if apply_update:
            self.apply_gradients(layers_done=early)
            if early:
                torch.cuda.current_stream(self.dev).wait_stream(self._opt_stream())     # the next forward reads the layers' new weights and packings
        return metrics

    grad_allreduce_dtype = 'f32'      # 'bf16': each range is all-reduced as a bf16 copy (177 MB instead of 354 MB on the links; the sum
                                      # of the replicas' gradients is then rounded to 8 bits per replica term — an option, off by default:
                                      # the reference's MirroredStrategy reduces fp32 variables' gradients in fp32)
    time_allreduce = False            # record HIP events around the wait for the collectives (exposed_allreduce_ms())
    _allreduce_events = None

    def _allreduce_range(self, a, b):
        """async SUM all-reduce of flat_g[a:b] -> (handle, bf16 staging buffer or None, (a, b))"""
        if self.grad_allreduce_dtype == 'bf16':
            buf = self.flat_g[a:b].to(torch.bfloat16)
            return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), buf, (a, b)
        return dist.all_reduce(self.flat_g[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, (a, b)

    def exposed_allreduce_ms(self):
        """time the compute stream spent waiting for the gradient collectives after the last backward kernel of the most recent step
        (needs ``time_allreduce = True``): what the per-layer overlap did not hide"""
        if self._allreduce_events is None:
            return None
        self._allreduce_events[1].synchronize()
        return self._allreduce_events[0].elapsed_time(self._allreduce_events[1])

    # ------------------------------------------------------------------ evaluation steps (Keras Model.evaluate / predict)
    def test_step(self, poses, tokens, codebook_model=None):
        """``MIGT.test_step`` (migt.py:507-527): the compute_losses graph with ``training=False``, the same loss / accuracy metrics as the
        training step, the pose-head errors, and — given the codebook model — the PSNR of the image decoded from the generated tokens of
        the LAST view against the image decoded from its ground-truth tokens (:521-526)."""
        metrics, out = self.train_step(poses, tokens, _forward_only=True)
        c, skip = self.cfg, self.cfg.n_loss_skip
        if out['pose_head_raw'] is not None:                                         # :514-519 on pose_prediction[:, n_loss_skip:]
            B, S = tokens.shape[:2]
            pp = geometry.pose_head_postprocess(out['pose_head_raw'], c.pose_multiplier)[:, skip:]
            gt = poses.to(self.dev).view(B, S, 1, 7)[:, skip:]
            # AllowNanMean (utils/metrics.py:75-87) as the reference computes it: a NaN sample — asin of a norm that rounding pushed past 1,
            # a zero quaternion — is REPLACED BY 0 and still counted (the weight line :86 tests the already-cleaned values, so no sample is
            # ever dropped): mean over all samples of nan_to_zero(value).  No clamp on the asin argument.
            nan_mean = lambda v: torch.where(torch.isnan(v), torch.zeros_like(v), v).mean()      # noqa: E731
            metrics['pose_pos_err'] = nan_mean((pp[..., :3] - gt[..., :3]).norm(dim=-1))         # CameraPositionError, :90-95
            q1 = geometry.quaternion_normalize(pp[..., 3:])                                     # CameraOrientationError, :98-110: the sine
            q2 = geometry.quaternion_normalize(gt[..., 3:].expand_as(pp[..., 3:]))              # form, stable near zero rotation
            diff = geometry.quaternion_multiply(q1, geometry.quaternion_conjugate(q2))
            metrics['pose_ori_err'] = nan_mean(2.0 * torch.asin(diff[..., 1:].norm(dim=-1)))
        if codebook_model is not None:
            t = c.token_image_size
            gen = out['predicted_tokens'][:, -1].reshape(-1, t, t)
            img = [(codebook_model.decode_code(x.to(torch.int64)).float() / 2 + 0.5).clamp(0, 1) for x in (gen, tokens[:, -1].to(self.dev))]
            mse = ((img[0] - img[1]) ** 2).reshape(img[0].shape[0], -1).mean(1)
            metrics['psnr'] = (10.0 * torch.log10(1.0 / mse.clamp_min(1e-12))).mean()    # tf.image.psnr(max_val=1), batch mean
        return metrics

    def predict_step(self, poses, tokens, codebook_model):
        """``MIGT.predict_step`` (migt.py:532-541): arg-max tokens of every view from the masked stream (special ids -> 0), decoded next
        to the decoded ground-truth tokens."""
        _, out = self.train_step(poses, tokens, _forward_only=True)
        t, nE = self.cfg.token_image_size, self.cfg.n_embeddings
        gen = out['predicted_tokens'].reshape(-1, tokens.shape[1], t, t)
        gen = torch.where(gen < nE, gen, torch.zeros_like(gen))
        dec = codebook_model.decode_code(gen.reshape(-1, t, t).to(torch.int64))
        gt = codebook_model.decode_code(tokens.to(self.dev).reshape(-1, t, t).to(torch.int64))
        return dict(decoded_image=dec, latent_code=gen, ground_truth_image=gt)

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def apply_gradients(self, layers_done: bool = False):
        """``layers_done``: the transformer layers' ranges were already updated (and re-packed) by train_step's early per-layer optimizer: only the
        head range (embeddings, pose heads, ln_f) is left"""
        c = self.cfg
        lr, lr_adam = self._adam_scalars()
        if self.fused_optimizer:
            # one launch over the flat buffer; the "bias" tensors (models/utils.py:424: the only names excluded) as no-decay ranges
            if self._nodecay is None:
                r = [[self.slices[n][0], self.slices[n][1]] for n in self.names if 'bias' in n]
                self._nodecay = torch.tensor(sorted(r), dtype=torch.int64, device=self.dev).reshape(-1, 2)
            if layers_done:
                a, b = self.head_range
                if self._head_nodecay is None:
                    r = [[self.slices[n][0], self.slices[n][1]] for n in self.names if 'bias' in n and self.slices[n][0] < b]
                    self._head_nodecay = torch.tensor(sorted(r), dtype=torch.int64, device=self.dev).reshape(-1, 2)
                T.adamw_flat_(self.flat_p[a:b], self.flat_g[a:b], self.flat_m[a:b], self.flat_v[a:b],
                              self._head_nodecay if c.weight_decay > 0 else None, lr * c.weight_decay if c.weight_decay > 0 else 0.0, lr_adam,
                              self.b1, self.b2, self.eps)
            elif self.fused_optimizer_repack and self._adam_pack is not None and self._pack16 is not None:
                T.adamw_flat_pack_(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self._nodecay if c.weight_decay > 0 else None,
                                   lr * c.weight_decay if c.weight_decay > 0 else 0.0, lr_adam, self.b1, self.b2, self.eps, self._adam_pack)
                self._packs_fresh = True
            else:
                T.adamw_flat_(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self._nodecay if c.weight_decay > 0 else None,
                              lr * c.weight_decay if c.weight_decay > 0 else 0.0, lr_adam, self.b1, self.b2, self.eps)
        else:
            for n in self.names:
                a, b, _ = self.slices[n]
                decay = c.weight_decay > 0 and 'bias' not in n            # models/utils.py:424: only "bias" names are excluded
                T.adamw_(self.flat_p[a:b], self.flat_g[a:b], self.flat_m[a:b], self.flat_v[a:b],
                         lr * c.weight_decay if decay else 0.0, lr_adam, self.b1, self.b2, self.eps)
        self.step_count += 1
        self._early_layers_done = bool(layers_done)
        try:
            self.repack()
        finally:
            self._early_layers_done = False

    def state_dict(self):
        return {n: self.p(n).detach().cpu().clone() for n in self.names}

    # ------------------------------------------------------------------ resume / finetune (the optimizer half of a checkpoint)
    def optimizer_state_dict(self):
        """what ``model.load_weights(checkpoint)`` restores beside the weights (finetune_transformer.py:76-85: "this restores the model and
        the optimizer"): Adam's first / second moments per parameter, optimizer.iterations and the schedule's offset"""
        sd = {'iterations': int(self.step_count), 'lr_offset': int(self.lr_offset)}
        for n in self.names:
            a, b, s = self.slices[n]
            sd['m/' + n] = self.flat_m[a:b].view(s).detach().cpu().clone()
            sd['v/' + n] = self.flat_v[a:b].view(s).detach().cpu().clone()
        return sd

    def load_optimizer_state_dict(self, sd, strict: bool = True):
        want = {p + n for n in self.names for p in ('m/', 'v/')}
        have = {k for k in sd if k.startswith(('m/', 'v/'))}
        if strict and want != have:
            raise RuntimeError(f'optimizer state: missing {sorted(want - have)[:4]}, unexpected {sorted(have - want)[:4]}')
        for n in self.names:
            a, b, s = self.slices[n]
            for pre, flat in (('m/', self.flat_m), ('v/', self.flat_v)):
                if pre + n in sd:
                    t = torch.as_tensor(sd[pre + n], dtype=torch.float32)
                    if tuple(t.shape) != tuple(s):
                        raise RuntimeError(f'optimizer state {pre}{n}: shape {tuple(t.shape)} != {tuple(s)}')
                    flat[a:b] = t.reshape(-1).to(self.dev)
        self.step_count = int(sd.get('iterations', self.step_count))
        self.lr_offset = int(sd.get('lr_offset', self.lr_offset))
        return self

    def begin_finetune(self, learning_rate: float = 1e-5, total_steps: int = None, warmup_steps: int = 2000):
        """viewformer/train/finetune_transformer.py:76-86: a NEW schedule (the finetune learning rate — default 1e-5, :22 — over the finetune
        run's ``total_steps`` with 2000 warm-up steps) on the RESTORED optimizer (moments and iteration count kept: Adam's bias correction goes
        on from the restored count), with the schedule's origin moved to that count (``lr_schedule.offset.assign(optimizer.iterations)``).
        Call after load_state_dict / load_optimizer_state_dict; config overrides (pose_multiplier, localization_weight, ... :57-70) are made
        on the model's config before the trainer is built, as the reference makes them before load_model."""
        self.lr_init = float(learning_rate)          # (the model's config keeps its own total_steps: the localization-weight schedule, migt.py:268)
        if total_steps is not None:
            self.lr_total_steps = int(total_steps)
        self.warmup_steps = int(warmup_steps)
        self.lr_offset = int(self.step_count)
        return self


def finetune_transformer(checkpoint: str, total_steps: int, learning_rate: float = 1e-5, device='cuda', precision: str = 'f32',
                         process_group=None, **transformer_config) -> MIGTTrainer:
    """The model-and-optimizer setup of ``viewformer/train/finetune_transformer.py:57-86`` (the data loop and callbacks around it are the
    caller's): config overrides (pose_multiplier, localization_weight, sequence_size, n_loss_skip, weight_decay, gradient_clip_val,
    augment_poses — only those given) go into ``load_model`` BEFORE the model is built (:57-73); the optimizer is created with the finetune
    learning rate over the finetune run's ``total_steps`` and 2000 warm-up steps, weight decay from the loaded config (:78-82);
    ``model.load_weights(checkpoint).expect_partial()`` restores weights AND optimizer (:84); ``lr_schedule.offset`` is set to the restored
    iteration count (:85).  ``precision='bf16'`` is the driver's ``--fp16`` (:53-55)."""
    from . import checkpoint as ck
    overrides = {k: v for k, v in transformer_config.items() if v is not None}
    model = ck.load_model(checkpoint, device=None, **overrides)
    if not isinstance(model, MIGT):
        raise RuntimeError('finetune_transformer needs a transformer checkpoint')
    if precision != model.precision:
        model = MIGT(model.config, precision=precision).load_state_dict(model._sd_host)
    trainer = MIGTTrainer(model.to(device), process_group=process_group)
    if not checkpoint.endswith(('.pth', '.ckpt')):
        ck.restore_optimizer(trainer, checkpoint, strict=False)             # .expect_partial(): a weights-only checkpoint is accepted
    return trainer.begin_finetune(learning_rate=learning_rate, total_steps=total_steps, warmup_steps=2000)

