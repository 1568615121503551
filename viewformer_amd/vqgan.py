"""VQGAN codebook model on MI355X — host-side mirror of the reference's model object.

Mirrors the duck-typed protocol every reference caller uses:
``.config``, ``.encode(x) -> (quant, diff, codes[int64])``, ``.decode(quant)``,
``.decode_code(codes)``, ``.to(device)``, ``load_state_dict`` with the reference's key set
(viewformer/models/vqgan_th.py:321-398, utils_th.py:8-72).  Two calling conventions:
``data_format='NCHW'`` = the Torch model (vqgan_th.py), ``'NHWC'`` = its TF twin
(viewformer/models/vqgan.py:291-301) that the evaluators call.

All arithmetic runs in libvf_hip.so: channels-last fp32 activations, implicit-GEMM convolutions in one of the fp32-EQUIVALENT
arithmetics (``conv_arith`` = 'x3h' — two exact fp16 pieces, three MFMA products, the default — 'x6' or the native 'f32' MFMA;
DESIGN.md 3), GroupNorm statistics + apply fused into the producing / consuming convolution, the codebook lookup as an fp16-MFMA
candidate filter with an exact fp32 re-rank (``lookup`` = 'filter'; 'exact' scans every code on the f32 MFMA).
``decoder_precision='bf16'`` is the tolerance arm of the decoder.  torch is used for device allocations, views and the stream only.
"""
import re
from collections import OrderedDict

import os

import numpy as np
import torch

from . import ops
from .config import VQGANConfig
from .weights import vqgan_layout

_IGNORED = re.compile(r'(perceptual_loss\..*)|(loss\..*)')     # vqgan_th.py:322


class _Conv:
    __slots__ = ('wp', 'wp16', 'wp6', 'wp3h', 'bias', 'cin', 'cout', 'k', 'w_raw')


class VQGAN:
    def __init__(self, config: VQGANConfig = None, data_format: str = 'NCHW', device=None, max_images_per_call: int = 1024,
                 decoder_precision: str = 'f32', conv_arith: str = 'x3h', lookup: str = 'filter'):
        """``conv_arith`` picks the fp32-EQUIVALENT arithmetic of the convolutions: 'f32' = native f32 MFMA everywhere;
        'x6' = six exact bf16 partial products per fp32 product (csrc/conv3_halo_x6.hip, gemm_x6.hip); 'x3h' (default) = three exact
        fp16 partial products (csrc/conv3_halo_x3h.hip, gemm_x3h.hip, attn_spatial.hip) for every 3x3 convolution (stride 1, stride 2,
        nearest-x2), the 1x1 convolutions (quant_conv included) / q|k|v projections and the AttnBlock core, with x6 left for the
        shapes the x3h kernels do not tile.  Each has an error against fp64 no larger than the native f32 MFMA's and all give
        bit-identical token indices on the reference's golden vectors (tests/test_hip_models.py, test_hip_parity_scale.py).
        ``decoder_precision='bf16'`` runs the DECODER's wide 3x3 convolutions and 1x1 projections on the bf16-MFMA
        arm (decoded pixels are tolerance-bounded in the north star); the encoder and the codebook lookup are always
        exact fp32 so token indices stay bit-exact.  ``lookup``: 'filter' = fp16-MFMA candidate filter + exact fp32 re-rank
        (csrc/vq_filter.hip), 'exact' = every code on the f32 MFMA (csrc/vq_argmin.hip); identical indices."""
        self.config = config or VQGANConfig()
        assert data_format in ('NCHW', 'NHWC')
        assert decoder_precision in ('f32', 'bf16') and conv_arith in ('f32', 'x6', 'x3h') and lookup in ('filter', 'exact')
        self.lookup = lookup
        self.decoder_precision = decoder_precision
        self.conv_arith = conv_arith
        self.data_format = data_format
        self.device = torch.device(device) if device is not None else None
        self.max_images_per_call = max_images_per_call
        self._enc_plan, self._dec_plan = vqgan_layout(self.config)
        self._sd = None          # reference-keyed fp32 tensors on the device
        self._conv = {}
        self._norm = {}
        self._qkv = {}
        self.training = False
        # conv_arith='x3h': 1x1 convolutions / q|k|v projections on the 3-product split-fp16 GEMM (VF_VQ_DENSE_X3H=0: 6-product x6); read
        # per instance, like every other environment knob of the package
        self.dense_x3h = os.environ.get('VF_VQ_DENSE_X3H', '1') != '0'

    # ------------------------------------------------------------------ module-ish API
    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise ops._lib.VfError('viewformer_amd.VQGAN runs on the GPU only (no CPU fallback)')
        self.device = device
        if self._sd_host is not None:
            self._upload()
        return self

    _sd_host = None

    def expected_keys(self):
        keys = []
        for kind, name, args in self._enc_plan + self._dec_plan:
            if kind in ('conv3', 'down', 'up'):
                keys += [name + '.weight', name + '.bias']
            elif kind == 'res':
                for p in ('norm1', 'conv1', 'norm2', 'conv2'):
                    keys += [f'{name}.{p}.weight', f'{name}.{p}.bias']
                if args[0] != args[1]:
                    keys += [f'{name}.nin_shortcut.weight', f'{name}.nin_shortcut.bias']
            elif kind == 'attn':
                for p in ('norm', 'q', 'k', 'v', 'proj_out'):
                    keys += [f'{name}.{p}.weight', f'{name}.{p}.bias']
            elif kind == 'norm_swish':
                keys += [name + '.weight', name + '.bias']
        keys += ['quant_conv.weight', 'quant_conv.bias', 'post_quant_conv.weight', 'post_quant_conv.bias',
                 'quantize.embeddings', 'quantize.ema_cluster_size_hidden', 'quantize.ema_dw_hidden', 'quantize.counter']
        return keys

    def load_state_dict(self, state_dict, strict: bool = True):
        """Same contract as vqgan_th.py:346-359: metric/loss keys are ignored, otherwise
        missing / unexpected keys raise RuntimeError."""
        sd = {k: v for k, v in state_dict.items() if not _IGNORED.match(k)}
        if strict:
            want = set(self.expected_keys())
            have = set(sd.keys())
            if want - have:
                raise RuntimeError(f'Missing keys: {want - have}')
            if have - want:
                raise RuntimeError(f'Unexpected keys: {have - want}')
        host = OrderedDict()
        for k, v in sd.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            host[k] = np.ascontiguousarray(v)
        self._sd_host = host
        if self.device is not None:
            self._upload()
        return self

    def state_dict(self):
        return OrderedDict((k, torch.from_numpy(v)) for k, v in self._sd_host.items())

    # ------------------------------------------------------------------ weight upload / packing
    def _upload(self):
        dev = self.device
        h = self._sd_host
        self._conv, self._norm, self._qkv = {}, {}, {}
        self._act16_cache = {}

        def dev_t(name):
            return torch.from_numpy(h[name]).to(dev, torch.float32).contiguous()

        def conv(name):
            w = dev_t(name + '.weight')
            c = _Conv()
            c.cout, c.cin, c.k = w.shape[0], w.shape[1], w.shape[2]
            c.bias = dev_t(name + '.bias')
            c.w_raw = w
            c.wp = ops.pack_conv_oihw(w) if c.cin % 32 == 0 else None
            c.wp16 = None
            c.wp6 = None
            c.wp3h = None
            split = self.conv_arith in ('x6', 'x3h')
            if self.conv_arith == 'x3h' and c.k == 3 and c.cin == 3 and c.cout % 128 == 0:
                c.wp3h = ops.pack_conv_in_x3h(w)
            if split and c.k == 3 and c.cin % 32 == 0 and c.cout % 128 == 0:
                c.wp6 = ops.pack_conv3_x6(w)
                if self.conv_arith == 'x3h':
                    c.wp3h = ops.pack_conv3_x3h(w)
            elif split and c.k == 1 and c.cin % 64 == 0 and c.cout >= 64:
                c.wp6 = ops.pack_dense_nk_x6(w.reshape(c.cout, c.cin))
                if self.conv_arith == 'x3h' and self.dense_x3h:       # 1x1 convolutions on the 3-product GEMM too (csrc/gemm_x3h.hip)
                    c.wp3h = ops.pack_dense_nk_x3h(w.reshape(c.cout, c.cin))
            if self.decoder_precision == 'bf16' and (name.startswith('decoder.') or name == 'post_quant_conv'):
                if c.k == 3 and c.cin % 32 == 0 and c.cout % 128 == 0:
                    c.wp16 = ops.pack_conv3_bf16(w)
                elif c.k == 1 and c.cin % 64 == 0:
                    c.wp16 = ops.pack_dense_nk_bf16(w.reshape(c.cout, c.cin))
            self._conv[name] = c

        def norm(name):
            self._norm[name] = (dev_t(name + '.weight'), dev_t(name + '.bias'))

        for kind, name, args in self._enc_plan + self._dec_plan:
            if kind in ('conv3', 'down', 'up'):
                conv(name)
            elif kind == 'res':
                norm(name + '.norm1'); conv(name + '.conv1'); norm(name + '.norm2'); conv(name + '.conv2')
                if args[0] != args[1]:
                    conv(name + '.nin_shortcut')
            elif kind == 'attn':
                norm(name + '.norm')
                c = args[0]
                # fused q|k|v projection: one [C][3C] GEMM
                w = torch.cat([dev_t(f'{name}.{p}.weight').reshape(c, c) for p in ('q', 'k', 'v')], 0)   # [3C][C] (out,in)
                b = torch.cat([dev_t(f'{name}.{p}.bias') for p in ('q', 'k', 'v')], 0)
                if self.conv_arith == 'x3h' and self.dense_x3h and c % 64 == 0:
                    self._qkv[name] = (ops.pack_dense_nk_x3h(w), b)
                else:
                    self._qkv[name] = (ops.pack_dense_nk_x6(w) if self.conv_arith in ('x6', 'x3h') and c % 64 == 0 else ops.pack_dense_nk(w), b)
                conv(name + '.proj_out')
            elif kind == 'norm_swish':
                norm(name)
        conv('quant_conv')
        conv('post_quant_conv')
        self._E = dev_t('quantize.embeddings')                      # [D][Kc]
        self._E_packed, self._e_sq = ops.vq_pack_codebook(self._E)
        # the filtered lookup (fp16 candidate filter + exact re-rank: identical indices, ~10x less matrix time) where its shape
        # rules hold; the exact f32-MFMA kernel otherwise.  ``lookup='exact'`` forces the latter (tests cross-check the two).
        self._E_filter = (ops.vq_filter_pack(self._E) if self.lookup == 'filter' and ops.vq_filter_supported(*self._E.shape)
                          else None)
        torch.cuda.synchronize(dev)

    # ------------------------------------------------------------------ building blocks (NHWC rows)
    def _conv3(self, x, name, n, H, W, mode=ops.MODE_CONV3_S1, pro=None, pro_swish=True, res=None, o16=False):
        """``o16``: write the output as bf16 (bf16 arm, halo-kernel shapes only); a bf16 ``x`` is read as such.  The activation stream
        switches to bf16 once (at the 16 -> 32 upsample convolution, see _run_plan) and stays bf16 to conv_out."""
        c = self._conv[name]
        a16 = x.dtype == torch.bfloat16
        if mode == ops.MODE_CONV3_S2PAD:
            Ho, Wo = H // 2, W // 2
        elif mode == ops.MODE_CONV3_UP2:
            Ho, Wo = H * 2, W * 2
        else:
            Ho, Wo = H, W
        if res is None and ops.conv3_small_cout_supported(mode, c.cin, c.cout, Ho, Wo):      # conv_out (-> 3 channels)
            self._stats_of = None
            return ops.conv3_small_cout(x, c.w_raw, c.bias, n, H, W, c.cin, c.cout, pro=pro, pro_swish=pro_swish), Ho, Wo
        o16 = o16 or a16
        out = torch.empty((n * Ho * Wo, c.cout), dtype=torch.bfloat16 if o16 else torch.float32, device=x.device)
        # the 8x8 stage (512 channels, first 11 convs of the decoder) stays fp32 even in the bf16 arm: rounding there is
        # amplified by every later layer (8.3e-2 max pixel error with it in bf16 vs 4.6e-2 without) and it is ~1 ms of work
        bf16 = (c.wp16 is not None and mode in (ops.MODE_CONV3_S1, ops.MODE_CONV3_UP2) and Ho % 8 == 0 and Wo % 16 == 0)
        if o16 and not bf16:
            raise ops._lib.VfError(f'{name}: bf16 activations need the bf16 convolution arm (halo-kernel shape)')
        x3h = (not bf16 and c.wp3h is not None and ops.conv3_x3h_supported(mode, c.cin, c.cout, Ho, Wo)
               and not (mode == ops.MODE_CONV3_S2PAD and pro is not None))
        x6 = (not bf16 and not x3h and c.wp6 is not None and ops.conv3_x6_supported(mode, c.cin, c.cout, Ho, Wo)
              and not (mode == ops.MODE_CONV3_S2PAD and pro is not None))
        # the halo kernels also emit the GroupNorm partial statistics of what they store, so the consumer's norm
        # (_gn) only runs the tiny finalize instead of re-reading the activation
        part = None
        if (bf16 or x6 or x3h) and self.fuse_gn_stats and c.cout in (128, 256, 512, 1024):
            part = ops.new_gn_part(n, Ho, Wo, x.device)
        ops.igemm(x, c.wp16 if bf16 else c.wp3h if x3h else c.wp6 if x6 else c.wp, n * Ho * Wo, c.cin, c.cout, out, bias=c.bias, res=res,
                  mode=mode, pro=pro, pro_swish=pro_swish, Hin=H, Win=W, Hout=Ho, Wout=Wo, bf16=bf16, x6=x6, x3h=x3h, gn_part=part,
                  a16=a16, o16=o16)
        self._stats_of = (out, part) if part is not None else None
        return out, Ho, Wo

    def _conv1(self, x, name, M, pro=None, pro_swish=False, rows_per_img=0, res=None):
        c = self._conv[name]
        a16 = x.dtype == torch.bfloat16           # a bf16 activation stream (nin_shortcut of the 64x64 level): bf16 in, bf16 out
        out = torch.empty((M, c.cout), dtype=x.dtype, device=x.device)
        bf16 = c.wp16 is not None and pro is None
        if a16 and not (bf16 and res is None):
            raise ops._lib.VfError(f'{name}: bf16 activations need the plain bf16 GEMM')
        x3h = not bf16 and c.k == 1 and c.wp3h is not None
        x6 = not bf16 and not x3h and c.wp6 is not None
        ops.igemm(x, c.wp16 if bf16 else c.wp3h if x3h else c.wp6 if x6 else c.wp, M, c.cin, c.cout, out, bias=c.bias, res=res, pro=pro,
                  pro_swish=pro_swish, pro_rows_per_img=rows_per_img, bf16=bf16, x6=x6, x3h=x3h, a16=a16, o16=a16)
        return out

    def _gn(self, x, name, n, HW, C):
        gamma, beta = self._norm[name]
        if self._stats_of is not None and self._stats_of[0] is x:
            mean_c, scale_c = ops.groupnorm_finalize(self._stats_of[1], gamma, n, HW, C, 32, 1e-6)
        else:
            if x.dtype != torch.float32:
                raise ops._lib.VfError(f'{name}: a bf16 activation carries its GroupNorm statistics from the producing kernel (fuse_gn_stats)')
            mean_c, scale_c = ops.groupnorm_stats(x, gamma, n, HW, C, 32, 1e-6)
        return (mean_c, scale_c, beta)

    _stats_of = None          # (tensor, partials) of the most recent halo-conv output
    # bf16 arm only: R > 0 = bf16 activations between the decoder's layers from resolution R x R up (see _run_plan).  Default 128: the last
    # level holds 6 of the decoder's 9.5 ms and costs one uint8 level of the stated bound (max 7 -> 8, mean |err| 4.4e-3 -> 4.6e-3;
    # R = 32: max 10, 5.3e-3 — tests/test_hip_bf16.py); 0 = fp32 activations throughout (the round-3 form)
    decoder_act16 = 128
    fuse_gn_stats = True
    fused_attention = True    # AttnBlock core in one kernel where the shape allows (False: batched GEMMs + row softmax, kept for A/B)

    def _res(self, x, name, n, H, W, cin, cout):
        """ResnetBlock.forward, vqgan_th.py:78-90"""
        p1 = self._gn(x, name + '.norm1', n, H * W, cin)
        h, _, _ = self._conv3(x, name + '.conv1', n, H, W, pro=p1)
        p2 = self._gn(h, name + '.norm2', n, H * W, cout)
        sc = x if cin == cout else self._conv1(x, name + '.nin_shortcut', n * H * W)
        out, _, _ = self._conv3(h, name + '.conv2', n, H, W, pro=p2, res=sc)
        return out

    def _attn(self, x, name, n, H, W, C):
        """AttnBlock.forward, vqgan_th.py:120-144"""
        HW = H * W
        M = n * HW
        pro = self._gn(x, name + '.norm', n, HW, C)
        wp, b = self._qkv[name]
        qkv = torch.empty((M, 3 * C), dtype=torch.float32, device=x.device)
        ops.igemm(x, wp, M, C, 3 * C, qkv, bias=b, pro=pro, pro_swish=False, pro_rows_per_img=HW, x6=wp.dtype == torch.bfloat16,
                  x3h=wp.dtype == torch.float16)
        if self.fused_attention and ops.attn_spatial_supported(HW, C):
            # scores, softmax and p.v in one kernel: the [HW][HW] matrix never leaves the CU (csrc/attn_spatial.hip)
            a = ops.attn_spatial(qkv, n, HW, C, float(int(C) ** (-0.5)), x3h=self.conv_arith == 'x3h' and self.dense_x3h)
            return self._conv1(a, name + '.proj_out', M, res=x)
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        # scores[b] = q_b @ k_b^T : B[kk=c][nn=key] = k[key][c]
        kp = ops.pack(k, C, HW, 1, sk=1, sn=3 * C, st=0, batch=n, src_bstride=HW * 3 * C)
        S = torch.empty((n, HW, HW), dtype=torch.float32, device=x.device)
        ops.igemm(q, kp, HW, C, HW, S, lda=3 * C, batch=n, stride_x=HW * 3 * C,
                  stride_w=ops.packed_floats(C, HW), stride_out=HW * HW)
        ops.softmax_rows_(S, n * HW, HW, float(int(C) ** (-0.5)))
        # out[b] = P_b @ v_b : B[kk=key][nn=c] = v[key][c]
        vp = ops.pack(v, HW, C, 1, sk=3 * C, sn=1, st=0, batch=n, src_bstride=HW * 3 * C)
        a = torch.empty((M, C), dtype=torch.float32, device=x.device)
        ops.igemm(S, vp, HW, HW, C, a, lda=HW, batch=n, stride_x=HW * HW, stride_w=ops.packed_floats(HW, C),
                  stride_out=HW * C)
        return self._conv1(a, name + '.proj_out', M, res=x)

    def _act16_plan_ok(self, plan, idx, H, W):
        """can the activation stream switch to bf16 at the 'up' convolution ``plan[idx]`` (output H x W)?  Only if EVERY layer from there to
        conv_out can take it: each 3x3 convolution on the bf16 halo arm (``wp16``: cin % 32 == 0, cout % 128 == 0, tile-aligned output) or
        the small-cout kernel (conv_out), each GroupNorm fed by fused partial statistics (producer cout in 128 / 256 / 512 / 1024), each
        nin_shortcut on the plain bf16 GEMM, no AttnBlock in the bf16 stretch.  Otherwise (e.g. ch = 64 or 96: a 64- or 384-channel top
        level) the decoder keeps fp32 activations and the per-layer choice of arithmetic, as before decoder_act16 existed."""
        key = (id(plan), idx, H, W)
        hit = self._act16_cache.get(key)
        if hit is not None:
            return hit

        def conv_ok(name, mode, h, w, feeds_norm=True):
            c = self._conv[name]
            ho, wo = (h * 2, w * 2) if mode == ops.MODE_CONV3_UP2 else (h, w)
            if ops.conv3_small_cout_supported(mode, c.cin, c.cout, ho, wo):
                return True                                              # conv_out: reads bf16, writes the fp32 image
            return (c.wp16 is not None and ho % 8 == 0 and wo % 16 == 0 and (not feeds_norm or c.cout in (128, 256, 512, 1024)))

        ok = self.fuse_gn_stats and conv_ok(plan[idx][1], ops.MODE_CONV3_UP2, H // 2, W // 2)
        h, w = H, W
        for kind, name, args in plan[idx + 1:]:
            if not ok:
                break
            if kind == 'res':
                ok = conv_ok(name + '.conv1', ops.MODE_CONV3_S1, h, w) and conv_ok(name + '.conv2', ops.MODE_CONV3_S1, h, w)
                if ok and args[0] != args[1]:
                    sc = self._conv[name + '.nin_shortcut']
                    ok = sc.wp16 is not None and sc.k == 1
            elif kind == 'up':
                ok = conv_ok(name, ops.MODE_CONV3_UP2, h, w)
                h, w = h * 2, w * 2
            elif kind == 'conv3':
                ok = conv_ok(name, ops.MODE_CONV3_S1, h, w, feeds_norm=False)
            elif kind == 'norm_swish':
                ok = True                                                # statistics come from the producing convolution (checked there)
            else:                                                        # 'attn' (fp32 q | k | v), 'down': no bf16-activation form
                ok = False
        self._act16_cache[key] = bool(ok)
        return bool(ok)

    def _run_plan(self, plan, x, n, H, W, first_is_image=False):
        for idx, (kind, name, args) in enumerate(plan):
            if kind == 'conv3':
                c = self._conv[name]
                if c.wp is None:      # 3-channel conv_in (fused uint8 -> [-1,1])
                    if c.wp3h is not None and ops.conv_in_x3h_supported(H, W, c.cout):      # matrix-pipe form + fused GN statistics
                        part = ops.new_gn_part(n, H, W, x.device) if self.fuse_gn_stats and c.cout in (128, 256, 512, 1024) else None
                        x = ops.conv_in(x, c.w_raw, c.bias, n, H, W, c.cout, wp3h=c.wp3h, gn_part=part).view(n * H * W, c.cout)
                        self._stats_of = (x, part) if part is not None else None
                    else:
                        x = ops.conv_in(x, c.w_raw, c.bias, n, H, W, c.cout).view(n * H * W, c.cout)
                        self._stats_of = None
                else:
                    x, H, W = self._conv3(x, name, n, H, W, pro=self._pending_pro, pro_swish=True)
                self._pending_pro = None
            elif kind == 'res':
                x = self._res(x, name, n, H, W, args[0], args[1])
            elif kind == 'attn':
                x = self._attn(x, name, n, H, W, args[0])
            elif kind == 'down':
                x, H, W = self._conv3(x, name, n, H, W, mode=ops.MODE_CONV3_S2PAD)
            elif kind == 'up':
                # decoder_act16 = R > 0: from the first upsample whose output is R x R or larger (R >= 32) the activations stay bf16 in HBM
                # to conv_out; the stages below keep fp32 activations
                o16 = bool(self.decoder_act16 and self.decoder_precision == 'bf16' and self.fuse_gn_stats and name.startswith('decoder.')
                           and H * 2 >= max(32, int(self.decoder_act16)) and (H * 2) % 8 == 0 and (W * 2) % 16 == 0
                           and x.dtype != torch.bfloat16 and self._act16_plan_ok(plan, idx, H * 2, W * 2))
                x, H, W = self._conv3(x, name, n, H, W, mode=ops.MODE_CONV3_UP2, o16=o16)
            elif kind == 'norm_swish':
                self._pending_pro = self._gn(x, name, n, H * W, args[0])   # folded into the next conv
        return x, H, W

    _pending_pro = None
    _act16_cache = None

    # ------------------------------------------------------------------ encoder / decoder on NHWC
    def _encode_nhwc(self, img):
        """img: [n,H,W,3] uint8 or float32 in [-1,1] -> (z rows [n*h*w, D], codes [n,h,w] int64)"""
        n, H, W, _ = img.shape
        x, h, w = self._run_plan(self._enc_plan, img.contiguous(), n, H, W)
        z = self._conv1(x, 'quant_conv', n * h * w)                     # vqgan_th.py:381
        cfg = self.config
        if self._E_filter is not None:
            codes = ops.vq_argmin_filtered(z, self._E_filter, cfg.embed_dim, cfg.n_embed)
        else:
            codes = ops.vq_argmin(z, self._E_packed, self._e_sq, cfg.embed_dim, cfg.n_embed)
        return z, codes.view(n, h, w)

    def _decode_rows(self, q_rows, n, h, w):
        """q_rows [n*h*w, D] -> NHWC float [n,H,W,out_ch]"""
        x = self._conv1(q_rows, 'post_quant_conv', n * h * w)            # vqgan_th.py:386
        x, H, W = self._run_plan(self._dec_plan, x, n, h, w)
        return x.view(n, H, W, self.config.out_ch)

    def _chunks(self, n):
        # a launch addresses its rows (images x pixels) with 32-bit counts: keep images * image_size^2 below 2^31
        m = max(1, min(self.max_images_per_call, (2 ** 31 - 1) // max(1, self.config.image_size ** 2)))
        return [(i, min(n, i + m)) for i in range(0, n, m)]

    # ------------------------------------------------------------------ public protocol
    def _to_nhwc(self, x):
        if x.dtype == torch.uint8:
            return x                                                    # evaluator entry: NHWC uint8
        return x.permute(0, 2, 3, 1) if self.data_format == 'NCHW' else x

    def encode(self, x):
        """VQGAN.encode (vqgan_th.py:379-383 / vqgan.py:291-295): -> (quant, diff, codes int64)."""
        self._require_ready()
        x = self._to_nhwc(x.to(self.device))
        n = x.shape[0]
        if n == 0:                                                        # an empty batch is not an error in the reference
            t = x.shape[1] // self.config.stride
            return _LazyEncodeResult(self, torch.empty((0, self.config.embed_dim), dtype=torch.float32, device=self.device),
                                     torch.empty((0, t, t), dtype=torch.int64, device=self.device))
        zs, cs = [], []
        for a, b in self._chunks(n):
            z, c = self._encode_nhwc(x[a:b])
            zs.append(z)
            cs.append(c)
        z = torch.cat(zs, 0) if len(zs) > 1 else zs[0]
        codes = torch.cat(cs, 0) if len(cs) > 1 else cs[0]
        return _LazyEncodeResult(self, z, codes)

    def encode_codes(self, x):
        """codes only (what every hot-path caller takes: ``encode(x)[-1]``)."""
        return self.encode(x)[-1]

    def decode_code(self, code_b):
        """VQGAN.decode_code (vqgan_th.py:390-393 / vqgan.py:297-301)."""
        self._require_ready()
        cfg = self.config
        codes = code_b.to(self.device).to(torch.int64).contiguous()
        n, h, w = codes.shape
        if n == 0:
            s = self.config.stride
            out = torch.empty((0, h * s, w * s, self.config.out_ch), dtype=torch.float32, device=self.device)
            return out.permute(0, 3, 1, 2) if self.data_format == 'NCHW' else out
        outs = []
        for a, b in self._chunks(n):
            q = ops.codebook_gather(self._E, codes[a:b], cfg.embed_dim, cfg.n_embed)   # embed_code, utils_th.py:70-72
            outs.append(self._decode_rows(q, b - a, h, w))
        out = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
        return out.permute(0, 3, 1, 2) if self.data_format == 'NCHW' else out

    def decode(self, quant):
        """VQGAN.decode (vqgan_th.py:385-388); quant in the model's data_format."""
        self._require_ready()
        q = quant.to(self.device)
        q = q.permute(0, 2, 3, 1) if self.data_format == 'NCHW' else q
        n, h, w, d = q.shape
        out = self._decode_rows(q.contiguous().view(n * h * w, d), n, h, w)
        return out.permute(0, 3, 1, 2) if self.data_format == 'NCHW' else out

    def __call__(self, x):
        """VQGAN.forward (vqgan_th.py:395-398): (dec, diff, quant, codes)."""
        quant, diff, codes = self.encode(x)
        return self.decode(quant), diff, quant, codes

    forward = __call__

    def _require_ready(self):
        if self._sd_host is None:
            raise RuntimeError('VQGAN: load_state_dict() first')
        if self.device is None:
            raise RuntimeError('VQGAN: .to("cuda") first (GPU only)')


class _LazyEncodeResult(tuple):
    """``(quant, diff, codes)`` — indexable like the reference's tuple; ``quant``/``diff`` are
    materialised only when actually read (hot-path callers take ``[-1]``)."""

    def __new__(cls, model, z, codes):
        self = super().__new__(cls, (None, None, codes))
        self._model, self._z, self._codes = model, z, codes
        self._qd = None
        return self

    def _quant_diff(self):
        if self._qd is None:
            m, cfg = self._model, self._model.config
            n, h, w = self._codes.shape
            if n == 0:
                quant = torch.empty((0, h, w, cfg.embed_dim), dtype=torch.float32, device=self._codes.device)
                self._qd = (quant.permute(0, 3, 1, 2) if m.data_format == 'NCHW' else quant,
                            torch.full((), float('nan'), device=self._codes.device))       # mean over nothing, as torch gives
                return self._qd
            q = ops.codebook_gather(m._E, self._codes, cfg.embed_dim, cfg.n_embed)
            z = self._z
            diff = (q - z).pow(2).mean()                 # utils_th.py:66  (reporting value, not on the hot path)
            quant = (z + (q - z)).view(n, h, w, cfg.embed_dim)          # straight-through value, utils_th.py:67
            if m.data_format == 'NCHW':
                quant = quant.permute(0, 3, 1, 2)
            self._qd = (quant, diff)
        return self._qd

    def __getitem__(self, i):
        if isinstance(i, slice):
            return tuple(self[j] for j in range(*i.indices(3)))
        i = i if i >= 0 else 3 + i
        if i == 2:
            return self._codes
        return self._quant_diff()[i]

    def __iter__(self):
        yield self[0]
        yield self[1]
        yield self[2]
