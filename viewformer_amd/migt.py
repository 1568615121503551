"""MIGT image-token transformer on MI355X — host-side mirror of the reference model object.

Protocol (reference viewformer/models/migt.py:241-455):
``model(dict(input_ids=[B,S,t,t] int, poses=[B,S or S-1,7] f32), training=False)
  -> dict(logits=[B,S,t,t,n_embeddings] f32, pose_prediction=[B,S,L,7] (if use_localization),
          hidden_states=[...])``, attributes ``mask_token``, ``use_localization``, ``config``,
``reduce_cameras(x, axis)``.  Inference graph only in this file (single stream:
evaluate_transformer.py:119-123,134-136); the multi-stream training/multictx graph is a
"next" row.

All arithmetic runs in libvf_hip.so: embedding-sum, LayerNorm, fused c_attn GEMM whose V|Q|K
thirds are read in place by the block-causal attention kernel, c_proj/MLP GEMMs with fused
bias / exact-erf GELU / residual epilogues, tied LM head.  torch = memory + stream only
(plus O(B*S) pose bookkeeping on [B,S,7] tensors).
"""
from collections import OrderedDict

import os

import numpy as np
import torch

from . import ops
from . import geometry
from .config import MIGTConfig


class _Dense:
    __slots__ = ('wp', 'wp16', 'wp6', 'bias', 'k', 'n', 'w_raw')


class MIGT:
    def __init__(self, config: MIGTConfig = None, device=None, skip_masked: bool = True, precision: str = 'f32',
                 dense_arith: str = 'x3h', bf16_activations: bool = True, attention: str = None):
        """``dense_arith`` picks how the fp32 dense layers are evaluated (precision='f32' only): 'f32' = native f32 MFMA,
        'x6' = the fp32-EQUIVALENT six-term split-bf16 GEMM (csrc/gemm_x6.hip: same error against fp64, ~1.8x faster),
        'x3h' = the three-term split-fp16 GEMM (csrc/gemm_x3h.hip: same error for activations in fp16's range — LayerNorm / GELU /
        attention outputs are —, half the matrix instructions; attention stays x6).
        ``precision='bf16'``: the dense layers (c_attn, c_proj, MLP, LM head, pose MLPs) run on the bf16-MFMA arm with
        fp32 activations / accumulation, and the attention contractions (q.k^T, p.v) on bf16 MFMA with an fp32 softmax;
        LayerNorm, the residual stream and the arg-max stay fp32.
        Logits are then tolerance-bounded (tests state the bound), as the north star allows for the transformer."""
        assert precision in ('f32', 'bf16') and dense_arith in ('f32', 'x6', 'x3h') and attention in (None, 'bf16', 'fp8')
        if attention is not None and precision != 'bf16':
            raise ValueError("attention='bf16' / 'fp8' belong to the tolerance arm: use precision='bf16'")
        # ``attention='fp8'`` (BASELINE configs[4]): q.k^T and p.v on v_mfma_f32_32x32x16_fp8_fp8 with OCP e4m3 operands, fp32 softmax
        # (csrc/attention_lp.hip); the dense layers stay on the bf16 arm.  Tolerances: tests/test_hip_fp8.py.
        self.attention = attention or ('bf16' if precision == 'bf16' else 'f32')
        self.precision = precision
        self.dense_arith = dense_arith
        # bf16 arm only: LayerNorm / GELU / attention outputs — values that only bf16 GEMMs consume — are written as bf16 by their
        # producers (exactly the rounding the GEMM would apply on load: bit-identical logits, half the traffic of the widest tensors)
        self.bf16_activations = bf16_activations
        self.config = config or MIGTConfig()
        c = self.config
        self.n_image_tokens = c.token_image_size ** 2
        self.mask_token = c.n_embeddings                      # migt.py:256
        self.localization_token = c.n_embeddings + 1          # migt.py:257
        self.use_localization = c.use_localization            # migt.py:269
        self.device = torch.device(device) if device is not None else None
        self.skip_masked = skip_masked
        self._sd_host = None
        self._dense = {}
        self._ln = {}

    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise ops._lib.VfError('viewformer_amd.MIGT runs on the GPU only (no CPU fallback)')
        self.device = device
        if self._sd_host is not None:
            self._upload()
        return self

    def expected_keys(self):
        c = self.config
        keys = ['wte.weight', 'wpe.embeddings']
        dense = ['pose_embedding.c_fc', 'pose_embedding.c_proj',
                 'pose_criterion.pose_classifier.c_fc', 'pose_criterion.pose_classifier.c_proj']
        lns = ['ln_f']
        for i in range(c.n_layer):
            dense += [f'h.{i}.attn.c_attn', f'h.{i}.attn.c_proj', f'h.{i}.mlp.c_fc', f'h.{i}.mlp.c_proj']
            lns += [f'h.{i}.ln_1', f'h.{i}.ln_2']
        for d in dense:
            keys += [d + '.weight', d + '.bias']
        for l in lns:
            keys += [l + '.gamma', l + '.beta']
        if c.use_dynamic_pose_loss:
            # DynamicLossWeightingCriterion (migt.py:105-112) is built whenever the flag is set (:279-280), localization head or not,
            # so a checkpoint of such a model always carries the pair; only the training step reads it
            keys.append('pose_loss_weighting_criterion.pos_ori_weights')
        return keys

    _DYN_KEY = 'pose_loss_weighting_criterion.pos_ori_weights'

    def load_state_dict(self, state_dict, strict: bool = True):
        # the one training-only variable (DynamicLossWeightingCriterion, migt.py:105-112, initial value [0, -3]) is tolerated either way:
        # state dicts written before the key was tracked lack it, and a Keras checkpoint of a model trained with the flag carries it
        # even when the loading config has the flag off (inference never reads it)
        if strict and (self._DYN_KEY in state_dict) != bool(self.config.use_dynamic_pose_loss):
            state_dict = OrderedDict(state_dict)
            if self.config.use_dynamic_pose_loss:
                state_dict[self._DYN_KEY] = np.array([0.0, -3.0], dtype=np.float32)
            else:
                del state_dict[self._DYN_KEY]
        if strict:
            want, have = set(self.expected_keys()), set(state_dict.keys())
            if want - have:
                raise RuntimeError(f'Missing keys: {want - have}')
            if have - want:
                raise RuntimeError(f'Unexpected keys: {have - want}')
        host = OrderedDict()
        for k, v in state_dict.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            host[k] = np.ascontiguousarray(v).reshape(-1) if k.endswith('.bias') else np.ascontiguousarray(v)
        self._sd_host = host
        if self.device is not None:
            self._upload()
        return self

    def state_dict(self):
        return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in self._sd_host.items())      # copies: host arrays may be read-only views

    def _upload(self):
        dev, h, c = self.device, self._sd_host, self.config

        def dev_t(name):
            return torch.from_numpy(h[name]).to(dev, torch.float32).contiguous()

        def dense(name, pack=True):
            d = _Dense()
            w = dev_t(name + '.weight')
            d.k, d.n = w.shape
            d.bias = dev_t(name + '.bias')
            d.w_raw = w
            d.wp = ops.pack_dense_kn(w) if (pack and d.k % 32 == 0) else None
            d.wp16 = ops.pack_dense_kn_bf16(w) if (self.precision == 'bf16' and pack and d.k % 64 == 0 and d.n >= 64) else None
            split = d.wp16 is None and self.dense_arith in ('x6', 'x3h') and pack and d.k % 64 == 0 and d.n >= 64
            # the packing's dtype names the arithmetic: bf16 planes = x6, f16 planes = x3h (_dense_launch)
            d.wp6 = (ops.pack_dense_kn_x3h(w) if self.dense_arith == 'x3h' else ops.pack_dense_kn_x6(w)) if split else None
            self._dense[name] = d

        def ln(name):
            self._ln[name] = (dev_t(name + '.gamma'), dev_t(name + '.beta'))

        self._wte = dev_t('wte.weight')
        self._wpe = dev_t('wpe.embeddings')
        self._lm_head = ops.pack_dense_nk(self._wte, n_rows=c.n_embeddings)    # logits sliced to n_embeddings (migt.py:417)
        self._lm_head16 = (ops.pack_dense_nk_bf16(self._wte, n_rows=c.n_embeddings)
                           if self.precision == 'bf16' and c.d_model % 64 == 0 else None)
        self._lm_head6 = ((ops.pack_dense_nk_x3h if self.dense_arith == 'x3h' else ops.pack_dense_nk_x6)(self._wte, n_rows=c.n_embeddings)
                          if self._lm_head16 is None and self.dense_arith in ('x6', 'x3h') and c.d_model % 64 == 0 else None)
        dense('pose_embedding.c_fc', pack=False)
        dense('pose_embedding.c_proj')
        dense('pose_criterion.pose_classifier.c_fc')
        dense('pose_criterion.pose_classifier.c_proj')
        for i in range(c.n_layer):
            ln(f'h.{i}.ln_1'); ln(f'h.{i}.ln_2')
            for p in ('attn.c_attn', 'attn.c_proj', 'mlp.c_fc', 'mlp.c_proj'):
                dense(f'h.{i}.{p}')
        ln('ln_f')
        torch.cuda.synchronize(dev)

    # ------------------------------------------------------------------ helpers
    def _gemm(self, x, name, M, epilogue=ops.EPI_NONE, res=None, out_bf16=False):
        d = self._dense[name]
        out = torch.empty((M, d.n), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
        self._dense_launch(x, d, M, out, res=res, epilogue=epilogue)
        return out

    @staticmethod
    def _dense_launch(x, d, M, out, res=None, epilogue=ops.EPI_NONE):
        """one dense layer on the arm its packing selects: bf16 (tolerance arm) > x6 (fp32-equivalent) > native f32 MFMA"""
        bf16, x6 = d.wp16 is not None, d.wp16 is None and getattr(d, 'wp6', None) is not None
        x3h = x6 and d.wp6.dtype == torch.float16
        ops.igemm(x, d.wp16 if bf16 else d.wp6 if x6 else d.wp, M, d.k, d.n, out, bias=d.bias, res=res, epilogue=epilogue,
                  bf16=bf16, x6=x6 and not x3h, x3h=x3h, a16=x.dtype == torch.bfloat16, o16=out.dtype == torch.bfloat16)

    def _lm(self, h, M, out):
        """tied LM head: logits = h @ wte[:n_embeddings]^T  (SharedEmbeddings._linear, migt.py:51-56,417)"""
        c = self.config
        bf16 = getattr(self, '_lm_head16', None) is not None
        x6 = not bf16 and getattr(self, '_lm_head6', None) is not None
        x3h = x6 and self._lm_head6.dtype == torch.float16
        ops.igemm(h, self._lm_head16 if bf16 else self._lm_head6 if x6 else self._lm_head, M, c.d_model, c.n_embeddings, out,
                  bf16=bf16, x6=x6 and not x3h, x3h=x3h)
        return out

    def _lm_argmax(self, h, M):
        """arg-max of the tied LM head WITHOUT materialising the logits (fused epilogue, csrc/lmhead_argmax.hip; bf16 arm): the codes
        ``tf.argmax(logits, -1)`` would give (evaluate_transformer.py:123), or None where the fused form does not apply (fp32 arm,
        odd shapes) — callers then compute the logits and call ops.argmax_rows."""
        c = self.config
        if getattr(self, '_lm_head16', None) is None or not ops.lmhead_argmax_supported(c.d_model, c.n_embeddings):
            return None
        return ops.lmhead_argmax_bf16(h, self._lm_head16, M, c.d_model, c.n_embeddings)

    def _pose_embed(self, poses):
        """pose_embedding(get_model_input(poses)) — migt.py:139-145,291,354 (fp32; multiplier 1 at inference)"""
        B, Sp, _ = poses.shape
        c = self.config
        pin = geometry.pose_model_input(poses, c.pose_multiplier).reshape(B * Sp, 7).contiguous()
        fc = self._dense['pose_embedding.c_fc']
        h1 = ops.dense_small_k(pin, fc.w_raw, fc.bias, B * Sp, 7, fc.n, gelu=True)
        return self._gemm(h1, 'pose_embedding.c_proj', B * Sp).view(B, Sp, c.d_model)

    def reduce_cameras(self, cameras, axis=-2):
        """MIGT.reduce_cameras, migt.py:532-533"""
        return geometry.reduce_cameras(cameras, axis)

    prune_last_block = True           # callers that read only the last views' hidden states (generate_and_localize: the MASK and LOC views) get the LAST
                                      # block's projection, LayerNorm, MLP and ln_f on those views' rows only: every earlier block needs all rows (they are the
                                      # keys and values of the next block), the last block's other rows feed nothing.  Row for row the same launches on fewer
                                      # rows (GEMM rows, LayerNorm rows are independent of each other): the rows that are returned keep their bits

    def _blocks(self, ids, add_emb, B, V, L, mask_spec=-1, tail_views=0):
        """embedding sum -> n_layer x Block -> ln_f over V views of L tokens.  ids [B,V,...] int,
        add_emb [B,V,d] (pose embedding or LOC-token row per view).  Returns [B*V*L, d]; with ``tail_views`` = k > 0 (and prune_last_block)
        the hidden states of the last k views only, [B*k*L, d]."""
        c, dev = self.config, self.device
        d, H = c.d_model, c.n_head
        T, M = V * L, B * V * L
        add = add_emb.contiguous().view(B * V, d)
        ids32 = ids.reshape(M).to(torch.int32).contiguous()
        h = ops.embed_sum(ids32, self._wte, self._wpe, add, B * V, L, d, c.n_embeddings + 2)   # migt.py:392
        l0 = self._dense['h.0.mlp.c_proj'] if c.n_layer else None
        act16 = (self.precision == 'bf16' and self.bf16_activations and d % 128 == 0 and l0 is not None and l0.k % 128 == 0
                 and all(self._dense[f'h.{i}.{n}'].wp16 is not None for i in range(c.n_layer)
                         for n in ('attn.c_attn', 'attn.c_proj', 'mlp.c_fc', 'mlp.c_proj')))
        att = torch.empty((M, d), dtype=torch.bfloat16 if act16 else torch.float32, device=dev)
        # the fused c_attn output is bf16 too (bit-identical downstream: the attention kernel rounds fp32 q/k/v to bf16 on load): half
        # the bytes written by the GEMM and read by the attention, whose LDS-DMA kernel (attention_dma.hip) takes bf16 tiles straight
        # into LDS.  VF_QKV16=0 keeps fp32 q/k/v (A/B runs).
        # (the fp8 arm's kernel stages its tiles through registers and takes fp32 q/k/v at full speed: no bf16 there)
        qkv16 = act16 and self.attention != 'fp8' and os.environ.get('VF_QKV16', '1') != '0'
        qkv = torch.empty((M, 3 * d), dtype=torch.bfloat16 if qkv16 else torch.float32, device=dev)
        for i in range(c.n_layer):                                           # Block.call, migt.py:230-238
            p = f'h.{i}'
            a = ops.layernorm(h, *self._ln[p + '.ln_1'], M, d, out_bf16=act16)
            ca = self._dense[p + '.attn.c_attn']
            self._dense_launch(a, ca, M, qkv)
            # thirds are (V, Q, K): migt.py:207-213
            ops.attn_blockcausal(qkv[:, d:2 * d], qkv[:, 2 * d:], qkv[:, :d], att, B, H, T, L,
                                 3 * d, 3 * d, 3 * d, d, 1.0, self.skip_masked, mask_spec, bf16=self.precision == 'bf16',
                                 x6=self.precision == 'f32' and self.dense_arith in ('x6', 'x3h'), fp8=self.attention == 'fp8')
            att_i = att
            if i == c.n_layer - 1 and 0 < tail_views < V:
                # the last block: only the requested views' rows go on (their attention rows and residual rows, gathered; see prune_last_block)
                k = tail_views
                M = B * k * L
                att_i = att.view(B, V, L, d)[:, V - k:].reshape(M, d)
                h = h.view(B, V, L, d)[:, V - k:].reshape(M, d)
            h = self._gemm(att_i, p + '.attn.c_proj', M, res=h)
            m = ops.layernorm(h, *self._ln[p + '.ln_2'], M, d, out_bf16=act16)
            f = self._gemm(m, p + '.mlp.c_fc', M, epilogue=ops.EPI_GELU, out_bf16=act16)
            h = self._gemm(f, p + '.mlp.c_proj', M, res=h)
        if 0 < tail_views < V and c.n_layer == 0:
            h = h.view(B, V, L, d)[:, V - tail_views:].reshape(B * tail_views * L, d)
            M = B * tail_views * L
        return ops.layernorm(h, *self._ln['ln_f'], M, d)                    # migt.py:408

    def generate_and_localize(self, codes, cameras, codes_only: bool = False):
        """The evaluator's two transformer passes (evaluate_transformer.py:119-123 and :134-136) as ONE pass.

        Pass 1 feeds [codes[:, :-1], MASK] with all S poses; pass 2 feeds all S real code maps with S-1 poses
        and the LOC embedding on the last view.  Views 0..S-2 are identical inputs in both, and block-causal
        attention never lets them see the last view, so their hidden states are bit-identical: only the last
        view differs.  We run S+1 views [ctx_0..ctx_{S-2}, MASK-view, LOC-view] with the last two marked as
        twins (each attends to the context and itself, never to its sibling) — the reference's own branch-stream
        construction (migt.py:392-401, branching_attention.py:94-125).  Every produced row equals the two-pass
        row bit-for-bit (tests/test_hip_models.py), at 8/14 of the transformer work for S = 7.

        codes [B,S,t,t] int (all S views encoded), cameras [B,S,7] float32 (relative + normalised).
        Returns (logits_last [B,t,t,n_embeddings], pose_prediction_last [B,1,L,7]); with ``codes_only`` the first member is the
        generated code map [B,t,t] int64 instead (arg-max fused into the LM head where the arm supports it)."""
        if not self.use_localization:
            raise RuntimeError('generate_and_localize needs a model with the localization head')
        c, dev = self.config, self.device
        codes = codes.to(dev)
        cameras = cameras.to(dev)
        B, S = codes.shape[:2]
        tshape = tuple(codes.shape[2:])
        L = int(np.prod(tshape))
        d, nE = c.d_model, c.n_embeddings
        if d // c.n_head != 64:
            raise ops._lib.VfError('attention kernel supports head dim 64 only')
        pose_emb = self._pose_embed(cameras)                                            # [B,S,d]
        lpe = self._wte[self.localization_token].view(1, 1, d).expand(B, 1, d)
        add = torch.cat([pose_emb, lpe], 1)                                             # [B,S+1,d]
        mask = torch.full_like(codes[:, :1], self.mask_token)
        ids = torch.cat([codes[:, :-1], mask, codes[:, -1:]], 1)                        # [B,S+1,t,t]
        if self.prune_last_block:
            hf = self._blocks(ids, add, B, S + 1, L, mask_spec=S - 1, tail_views=2).view(B, 2, L, d)      # the MASK view and the LOC view
            h_mask = hf[:, 0].contiguous().view(B * L, d)
            h_loc = hf[:, 1].contiguous().view(B * L, d)
        else:
            hf = self._blocks(ids, add, B, S + 1, L, mask_spec=S - 1).view(B, S + 1, L, d)
            h_mask = hf[:, S - 1].contiguous().view(B * L, d)
            h_loc = hf[:, S].contiguous().view(B * L, d)
        gen = self._lm_argmax(h_mask, B * L) if codes_only else None
        if gen is None:
            lg = torch.empty((B * L, nE), dtype=torch.float32, device=dev)
            self._lm(h_mask, B * L, lg)                                                 # migt.py:417
            gen = ops.argmax_rows(lg, B * L, nE) if codes_only else None
        p1 = self._gemm(h_loc, 'pose_criterion.pose_classifier.c_fc', B * L, epilogue=ops.EPI_GELU)
        p2 = self._gemm(p1, 'pose_criterion.pose_classifier.c_proj', B * L)
        pose = geometry.pose_head_postprocess(p2.view(B, 1, L, 7), c.pose_multiplier)
        return (gen.view(B, *tshape) if codes_only else lg.view(B, *tshape, nE)), pose

    def _call_streams(self, ids, pose_emb, loc_tokens, out_poses, B, S, L, orig_shape):
        """Multi-stream ("branching") forward, migt.py:371-455 with training=False.

        Streams are laid out as extra views of one sequence (view index = stream*S + position) and the attention
        kernel's STREAMS mask gives a branch position i the main views < i plus its own tile — i.e.
        compute_causal_block_multiend_attention for every stream in one launch per layer.  Weights are shared
        across streams exactly as in Block.call (migt.py:230-238)."""
        c, dev = self.config, self.device
        d, nE = c.d_model, c.n_embeddings
        ids_streams = [ids.reshape(B, S, L)]
        add_streams = [pose_emb]
        img_ptr = pose_ptr = 0
        if out_poses is not None:                                            # migt.py:382-385,393-396
            out_poses = out_poses.to(dev)
            if out_poses.shape[1] != S:
                raise ValueError('output_poses must have one pose per view')
            ids_streams.append(torch.full((B, S, L), self.mask_token, dtype=ids.dtype, device=dev))
            add_streams.append(self._pose_embed(out_poses))
            img_ptr = len(ids_streams) - 1
        if loc_tokens is not None:                                           # migt.py:378-381,398-401
            lt = loc_tokens.to(dev).reshape(B, -1, L)
            if lt.shape[1] != S:
                raise ValueError('localization_tokens must have one token map per view')
            ids_streams.append(lt.to(ids.dtype))
            add_streams.append(self._wte[self.localization_token].view(1, 1, d).expand(B, S, d))
            pose_ptr = len(ids_streams) - 1
        NS = len(ids_streams)
        if NS > 1 and S == 1:
            # one view per stream: a branch position 0 sees no main view (j < 0) and its own tile, the main view sees itself — every
            # (scene, stream) is an independent 1-view sequence.  The kernels' STREAMS encoding (-S <= -2) cannot say S = 1 (-1 is
            # plain block-causal, where the streams would see each other), so run them as B*NS scenes of one view: same rows exactly.
            hf = self._blocks(torch.cat(ids_streams, 1), torch.cat(add_streams, 1), B * NS, 1, L, mask_spec=-1).view(B, NS, S, L, d)
        else:
            hf = self._blocks(torch.cat(ids_streams, 1), torch.cat(add_streams, 1), B, NS * S, L,
                              mask_spec=(-S if NS > 1 else -1)).view(B, NS, S, L, d)
        out = dict(hidden_states=[hf[:, s] for s in range(NS)])
        M = B * S * L
        hi = hf[:, img_ptr].contiguous().view(M, d)
        lg = torch.empty((M, nE), dtype=torch.float32, device=dev)
        self._lm(hi, M, lg)                                                  # migt.py:417
        out['logits'] = lg.view(*orig_shape, nE)
        if self.use_localization:                                            # migt.py:430-451
            hp = hf[:, pose_ptr].contiguous().view(M, d)
            p1 = self._gemm(hp, 'pose_criterion.pose_classifier.c_fc', M, epilogue=ops.EPI_GELU)
            p2 = self._gemm(p1, 'pose_criterion.pose_classifier.c_proj', M)
            out['pose_prediction'] = geometry.pose_head_postprocess(p2.view(B, S, L, 7), c.pose_multiplier)
        out['loss'] = 0
        return out

    # ------------------------------------------------------------------ forward (single stream, inference)
    def __call__(self, inputs, training=False, compute_losses=False, last_view_logits_only=False, last_view_codes_only=False):
        if training:
            raise RuntimeError('the training graph (dropout, losses, backward, optimizer: MIGT.train_step migt.py:464-505) is '
                               'viewformer_amd.train.MIGTTrainer(model).train_step(poses, tokens); __call__ is inference')
        if self._sd_host is None or self.device is None:
            raise RuntimeError('MIGT: load_state_dict() and .to("cuda") first')
        c, dev = self.config, self.device
        ids = inputs['input_ids'].to(dev)
        poses = inputs['poses'].to(dev)
        if poses.dtype != torch.float32:
            raise TypeError('poses must be float32 (migt.py:346)')
        orig_shape = tuple(ids.shape)
        B, S = orig_shape[:2]
        L = int(np.prod(orig_shape[2:]))
        d, H = c.d_model, c.n_head
        M = B * S * L
        if d // H != 64:
            raise ops._lib.VfError('attention kernel supports head dim 64 only')

        pose_emb = self._pose_embed(poses)                                   # [B,Sp,d]
        Sp = pose_emb.shape[1]
        if self.use_localization and S - Sp > 0:                             # migt.py:387-390
            lpe = self._wte[self.localization_token].view(1, 1, d).expand(B, S - Sp, d)
            pose_emb = torch.cat([pose_emb, lpe], 1)
        elif Sp != S:
            raise ValueError(f'poses has {Sp} views but input_ids has {S}')

        # ---- optional branch streams (migt.py:371-401): MASK stream for images, LOC stream for poses ----------
        loc_tokens = inputs.get('localization_tokens')
        out_poses = inputs.get('output_poses')
        if compute_losses:                                                   # forward graph of the training step
            if loc_tokens is None and self.use_localization:
                loc_tokens = ids
            if out_poses is None:
                out_poses = poses
        if loc_tokens is not None or out_poses is not None:
            if last_view_logits_only or last_view_codes_only:
                raise ValueError('last_view_logits_only / last_view_codes_only apply to the single-stream call')
            return self._call_streams(ids, pose_emb, loc_tokens, out_poses, B, S, L, orig_shape)

        hf = self._blocks(ids, pose_emb, B, S, L)

        out = dict(hidden_states=[hf.view(B, S, L, d)])
        nE = c.n_embeddings
        if last_view_codes_only:
            last_view_logits_only = True
        if last_view_logits_only:
            hl = hf.view(B, S, L, d)[:, -1].contiguous().view(B * L, d)
            gen = self._lm_argmax(hl, B * L) if last_view_codes_only else None
            if gen is None:
                lg = torch.empty((B * L, nE), dtype=torch.float32, device=dev)
                self._lm(hl, B * L, lg)
                out['logits_last'] = lg.view(B, *orig_shape[2:], nE)
                if last_view_codes_only:
                    gen = ops.argmax_rows(lg, B * L, nE)
            if gen is not None:
                out['codes_last'] = gen.view(B, *orig_shape[2:])                  # argmax(logits, -1)[:, -1], evaluate_transformer.py:123
        else:
            lg = torch.empty((M, nE), dtype=torch.float32, device=dev)
            self._lm(hf, M, lg)                                              # migt.py:417,51-56
            out['logits'] = lg.view(*orig_shape, nE)
        if self.use_localization:                                            # migt.py:430-451
            if last_view_logits_only:      # pose head on the last view only: shape [B,1,L,7], so [:, -1:] still works
                hp, Mp, Sp_out = hf.view(B, S, L, d)[:, -1].contiguous().view(B * L, d), B * L, 1
            else:
                hp, Mp, Sp_out = hf, M, S
            p1 = self._gemm(hp, 'pose_criterion.pose_classifier.c_fc', Mp, epilogue=ops.EPI_GELU)
            p2 = self._gemm(p1, 'pose_criterion.pose_classifier.c_proj', Mp)
            out['pose_prediction'] = geometry.pose_head_postprocess(p2.view(B, Sp_out, L, 7), c.pose_multiplier)
        out['loss'] = 0
        return out
