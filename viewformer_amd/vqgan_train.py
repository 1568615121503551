"""Codebook (VQGAN) training step on the GPU — ``viewformer-cli train codebook``:
``VQGAN.training_step`` (viewformer/models/vqgan_th.py:370-377) = forward in training mode (:349-352, incl. the EMA codebook update of
``QuantizeEMA.forward`` utils_th.py:46-64) -> ``_compute_loss`` (:354-368: mean |x - xrec| + codebook_weight * commitment) ->
autograd -> ``torch.optim.Adam(lr, betas=(0.5, 0.9))`` (:427-429), gradients averaged over replicas as Lightning's DDP does.

All arithmetic is in libvf_hip.so.  The backward pass launches the FORWARD convolution / GEMM kernels on re-arranged operands:
  dX of a 3x3 conv   = the same conv with the 180-degree-rotated, channel-transposed weight (stride 2: on the zero-inserted dY;
                       nearest-x2 upsample: followed by 2x2 block sums);
  dW of a conv       = one GEMM per tap: (tap-shifted activation, channel-major: vf_gather_transpose_f32) x dY, split-K;
  dX / dW of 1x1     = plain GEMMs;
plus the HBM-bound helpers of csrc/vqgan_bwd.hip (GroupNorm+swish backward, softmax backward, L1).  torch moves data only (views,
flips, zero-insertion, concatenation) and issues the collectives.  First version: correct and deterministic, not tuned — the GroupNorm
prologues are materialised instead of fused, packings are rebuilt every call.

``perceptual_weight > 0`` (the reference default, 1.0) needs the LPIPS-VGG weights the reference downloads: pass them as
``lpips_state_dict`` (lpips.load_lpips_weights / make_lpips_weights); without them the trainer refuses instead of silently dropping
the term.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from . import train_ops as T
from .vq_train import QuantizeEMATrainer
from .vqgan import VQGAN

_BUFFERS = ('quantize.embeddings', 'quantize.ema_cluster_size_hidden', 'quantize.ema_dw_hidden', 'quantize.counter')


class VQGANTrainer:
    def __init__(self, model: VQGAN, lr: float = None, betas=(0.5, 0.9), eps: float = 1e-8, ema_decay: float = 0.99,
                 process_group=None, lpips_state_dict=None):
        cfg = model.config
        if cfg.perceptual_weight > 0 and lpips_state_dict is None:
            raise ValueError('perceptual_weight > 0 needs the LPIPS-VGG weights (lpips_state_dict=...; viewformer_amd.lpips.'
                             'load_lpips_weights reads the upstream .pth files) — or set perceptual_weight=0.0 in VQGANConfig')
        if model._sd_host is None or model.device is None:
            raise RuntimeError('load_state_dict() and .to("cuda") the model first')
        self.model, self.cfg, self.dev = model, cfg, model.device
        self.lr = float(cfg.learning_rate if lr is None else lr)
        self.b1, self.b2, self.eps = float(betas[0]), float(betas[1]), float(eps)
        self.group = process_group
        self.step_count = 0
        host = model._sd_host
        self.names = [k for k in host if k not in _BUFFERS]
        sizes = [int(np.prod(host[k].shape)) for k in self.names]
        offs = np.concatenate([[0], np.cumsum([(s + 3) // 4 * 4 for s in sizes])]).astype(np.int64)
        self.slices = {k: (int(offs[i]), int(offs[i]) + sizes[i], tuple(host[k].shape)) for i, k in enumerate(self.names)}
        total = int(offs[-1])
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.flat_v = torch.zeros_like(self.flat_p)
        for k in self.names:
            self.p(k).copy_(torch.from_numpy(np.ascontiguousarray(host[k])).to(self.dev))
        self.quantizer = QuantizeEMATrainer(torch.from_numpy(np.ascontiguousarray(host['quantize.embeddings'])).to(self.dev),
                                            decay=ema_decay, process_group=process_group,
                                            ema_cluster_size_hidden=torch.from_numpy(np.ascontiguousarray(host['quantize.ema_cluster_size_hidden'])).to(self.dev),
                                            ema_dw_hidden=torch.from_numpy(np.ascontiguousarray(host['quantize.ema_dw_hidden'])).to(self.dev),
                                            counter=int(np.asarray(host['quantize.counter']).reshape(-1)[0]))
        self._enc_plan, self._dec_plan = model._enc_plan, model._dec_plan
        # debug switch (off in the product): issue every GroupNorm backward THREE times on the same live inputs and compare the results bit for
        # bit on the spot.  The only transient this code base has shown — several PROCESSES time-sliced on one GPU, one of three identical
        # launches returns lanes 48-63 of the sum(dt * xhat) accumulator wrong (profiles/r5_gpu_sharing_transient.txt) — is then caught where it
        # happens; launch #0's result is what the step goes on with (as the product would), the event goes to `transient_events`
        self.debug_triple_groupnorm_bwd = False
        self.transient_events = []
        self.lpips = None
        if cfg.perceptual_weight > 0:
            from .lpips import LPIPS
            self.lpips = LPIPS(lpips_state_dict, self.dev)

    # ------------------------------------------------------------------ parameter / gradient views
    def p(self, name):
        a, b, shape = self.slices[name]
        return self.flat_p[a:b].view(shape)

    def g(self, name):
        a, b, shape = self.slices[name]
        return self.flat_g[a:b].view(shape)

    def state_dict(self):
        sd = {k: self.p(k).clone() for k in self.names}
        sd.update(self.quantizer.state_dict())
        return sd

    def sync_model(self):
        """push the trained weights back into the inference model (re-packs everything)"""
        self.model.load_state_dict({k: v.detach().cpu().numpy() for k, v in self.state_dict().items()})
        return self.model

    # ------------------------------------------------------------------ GEMM helpers
    @staticmethod
    def _pack_b(b_rows, K, N):
        """B operand [K x N] (row-major) packed once for any number of ``_gemm`` calls against it"""
        if K % 64 == 0:
            return ('x6', ops.pack_dense_kn_x6(b_rows))
        return ('f32', ops.pack(b_rows, K, N, 1, sk=N, sn=1, st=0))

    @staticmethod
    def _gemm(a, b, M, K, N):
        """[M x K] @ [K x N] -> [M x N]; ``b`` is a row-major device tensor or a ``_pack_b`` result.  fp32-equivalent split-bf16
        kernel (split-K when the output has few tiles and the reduction is long) when K % 64 == 0, the fp32 MFMA GEMM otherwise"""
        kind, bp = b if isinstance(b, tuple) else VQGANTrainer._pack_b(b, K, N)
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        if kind == 'x6':
            tiles = ((M + 127) // 128) * ((N + 127) // 128)
            splits = max(1, min(256, 512 // max(tiles, 1), K // 512))
            if splits > 1 and (M * N) % 4 == 0:
                ops.gemm_x6_splitk(a, bp, M, K, N, out, splits, accumulate=False)
            else:
                ops.igemm(a, bp, M, K, N, out, x6=True)
            return out
        Kp = (K + 31) // 32 * 32
        if Kp != K:
            a2 = torch.zeros((M, Kp), dtype=torch.float32, device=a.device)
            a2[:, :K] = a
            a = a2
        ops.igemm(a, bp, M, Kp, N, out)
        return out

    # ------------------------------------------------------------------ convolution forward / backward
    def _conv_launch(self, x, w, bias, n, H, W, mode, res=None, activations=False):
        """forward conv (3x3 modes or 1x1) of NHWC rows x with OIHW weight w; picks the fp32-equivalent kernels when the shape allows.
        ``activations=True``: x is a forward activation (O(1) magnitudes) -> the three-product split-fp16 kernel (x3h) where it
        applies; gradients (dX convolutions, ~1e-6) stay on x6, which has no range condition"""
        cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
        if k == 1:
            M = n * H * W
            out = torch.empty((M, cout), dtype=torch.float32, device=x.device)
            w2 = w.reshape(cout, cin)
            if cin % 64 == 0 and cout >= 64:
                ops.igemm(x, ops.pack_dense_nk_x6(w2), M, cin, cout, out, bias=bias, res=res, x6=True)
            else:
                ops.igemm(x, ops.pack_dense_nk(w2), M, cin, cout, out, bias=bias, res=res)
            return out, H, W
        Ho, Wo = (H // 2, W // 2) if mode == ops.MODE_CONV3_S2PAD else (H * 2, W * 2) if mode == ops.MODE_CONV3_UP2 else (H, W)
        if res is None and ops.conv3_small_cout_supported(mode, cin, cout, Ho, Wo):
            return ops.conv3_small_cout(x, w, bias, n, H, W, cin, cout), Ho, Wo
        out = torch.empty((n * Ho * Wo, cout), dtype=torch.float32, device=x.device)
        x6 = ops.conv3_x6_supported(mode, cin, cout, Ho, Wo)
        x3h = x6 and activations
        ops.igemm(x, ops.pack_conv3_x3h(w) if x3h else ops.pack_conv3_x6(w) if x6 else ops.pack_conv_oihw(w), n * Ho * Wo, cin, cout, out,
                  bias=bias, res=res, mode=mode, Hin=H, Win=W, Hout=Ho, Wout=Wo, x6=x6 and not x3h, x3h=x3h)
        return out, Ho, Wo

    def _conv_fw(self, name, x, n, H, W, mode=ops.MODE_CONV3_S1, res=None):
        y, Ho, Wo = self._conv_launch(x, self.p(name + '.weight'), self.p(name + '.bias'), n, H, W, mode, res=res, activations=True)
        return y, (name, x, n, H, W, Ho, Wo, mode)

    def _conv_bw(self, ctx, dy, need_dx=True):
        name, x, n, H, W, Ho, Wo, mode = ctx
        w = self.p(name + '.weight')
        cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
        P = n * Ho * Wo
        gw = self.g(name + '.weight')
        fused = k == 3 and cout % 4 == 0 and T.conv3_wgrad_supported(cin, n, Ho, Wo)
        if not fused:
            T.colsum(dy, self.g(name + '.bias'), P, cout, accumulate=True)
        if k == 1:
            xt = T.gather_transpose(x, 1, 1, P, cin, 1, P)                                 # [cin][P]
            T.add_(gw.view(cout, cin), self._gemm(xt, dy, cin, P, cout).t().contiguous())
            if not need_dx:
                return None
            return self._gemm(dy, w.reshape(cout, cin).contiguous(), P, cout, cin)       # dX = dY . W
        if fused:                                                                          # dW and db in one gathered GEMM
            dw9 = T.conv3_wgrad(x, dy, n, H, W, cin, Ho, Wo, cout, mode)
            T.add_(gw, dw9[:9 * cin].view(3, 3, cin, cout).permute(3, 2, 0, 1).contiguous())
            T.add_(self.g(name + '.bias'), dw9[9 * cin].contiguous())
            return self._conv_dx(w, dy, n, H, W, Ho, Wo, mode) if need_dx else None
        # ---- dW, general shapes: the nine tap-shifted, channel-major copies of the input, one GEMM
        if mode == ops.MODE_CONV3_UP2:                                                     # the conv saw the nearest-x2 upsampled input
            xin = x.view(n, H, W, cin).repeat_interleave(2, 1).repeat_interleave(2, 2).contiguous()
            Hi, Wi, stride, off = Ho, Wo, 1, -1
        elif mode == ops.MODE_CONV3_S2PAD:                                                 # pad (0,1,0,1), stride 2, no other padding
            xin, Hi, Wi, stride, off = x, H, W, 2, 0
        else:
            xin, Hi, Wi, stride, off = x, H, W, 1, -1
        # all nine taps as one GEMM: [9*cin][P] (tap-shifted, channel-major) x dY [P][cout]
        xt = torch.empty((9, cin, P), dtype=torch.float32, device=dy.device)
        for ky in range(3):
            for kx in range(3):
                T.gather_transpose(xin, n, Hi, Wi, cin, Ho, Wo, stride, ky + off, kx + off, out=xt[ky * 3 + kx])
        dw = self._gemm(xt.view(9 * cin, P), dy, 9 * cin, P, cout)                        # [9*cin][cout]
        del xt
        dw = dw.view(3, 3, cin, cout).permute(3, 2, 0, 1).contiguous()                     # -> OIHW
        T.add_(gw, dw)
        return self._conv_dx(w, dy, n, H, W, Ho, Wo, mode) if need_dx else None

    def _conv_dx(self, w, dy, n, H, W, Ho, Wo, mode):
        """dX of a 3x3 convolution: the same convolution with the rotated, channel-transposed weight"""
        cout, cin = w.shape[0], w.shape[1]
        wr = w.flip(2, 3).permute(1, 0, 2, 3).contiguous()                                # [cin][cout][3][3] as an OIHW weight
        if cout % 32:                                                                      # conv_out (3 channels): pad the reduction dim
            cp = (cout + 31) // 32 * 32
            dyp = torch.zeros((dy.shape[0], cp), dtype=torch.float32, device=dy.device)
            dyp[:, :cout] = dy
            wrp = torch.zeros((cin, cp, 3, 3), dtype=torch.float32, device=dy.device)
            wrp[:, :cout] = wr
            dy, wr = dyp, wrp
        if mode == ops.MODE_CONV3_S2PAD:
            z = torch.zeros((n, H, W, dy.shape[1]), dtype=torch.float32, device=dy.device)  # dY at the odd positions of the input grid
            z[:, 1::2, 1::2] = dy.view(n, Ho, Wo, -1)
            dx, _, _ = self._conv_launch(z.view(n * H * W, -1), wr, None, n, H, W, ops.MODE_CONV3_S1)
            return dx
        du, _, _ = self._conv_launch(dy, wr, None, n, Ho, Wo, ops.MODE_CONV3_S1)
        if mode == ops.MODE_CONV3_UP2:
            return T.upsample2_bwd(du, n, H, W, cin)
        return du

    # ------------------------------------------------------------------ GroupNorm (+swish)
    def _gn_fw(self, name, x, n, HW, C, swish):
        gamma, beta = self.p(name + '.weight'), self.p(name + '.bias')
        mean_c, scale_c = ops.groupnorm_stats(x, gamma, n, HW, C, 32, 1e-6)
        a = ops.groupnorm_apply(x, mean_c, scale_c, beta, n, HW, C, swish=swish)
        return a, (name, x, mean_c, scale_c, n, HW, C, swish)

    def _gn_bw(self, ctx, da):
        name, x, mean_c, scale_c, n, HW, C, swish = ctx
        dx, dgamma, dbeta = T.groupnorm_bwd(x, da, mean_c, scale_c, self.p(name + '.weight'), self.p(name + '.bias'), n, HW, C, swish)
        if self.debug_triple_groupnorm_bwd:
            self._check_gn_bw_repeat(ctx, da, (dx, dgamma, dbeta))
        T.add_(self.g(name + '.weight'), dgamma)
        T.add_(self.g(name + '.bias'), dbeta)
        return dx

    def _check_gn_bw_repeat(self, ctx, da, first):
        name, x, mean_c, scale_c, n, HW, C, swish = ctx
        outs = [tuple(t.clone() for t in first)]
        for _ in range(2):
            outs.append(T.groupnorm_bwd(x, da, mean_c, scale_c, self.p(name + '.weight'), self.p(name + '.bias'), n, HW, C, swish))
        same = {(a, b): all(torch.equal(u, v) for u, v in zip(outs[a], outs[b])) for a, b in ((0, 1), (0, 2), (1, 2))}
        if same[0, 1] and same[0, 2]:
            return
        odd = 0 if same[1, 2] else 1 if same[0, 2] else 2 if same[0, 1] else -1        # -1: all three differ
        bad, good = outs[max(odd, 0)], outs[(max(odd, 0) + 1) % 3]
        ev = dict(layer=name, step=self.step_count, outlier_launch=odd, C=C, HW=HW, n_img=n, swish=bool(swish), differing={})
        for nm, a, b in zip(('dx', 'dgamma', 'dbeta'), bad, good):
            if not torch.equal(a, b):
                d = (a - b).abs()
                if nm == 'dx':
                    dd = d.view(n, HW, C)
                    ev['differing'][nm] = dict(images=dd.amax((1, 2)).nonzero().flatten().tolist(),
                                               channels=dd.amax((0, 1)).nonzero().flatten().tolist(), max_abs=float(d.max()))
                else:
                    ev['differing'][nm] = dict(channels=d.nonzero().flatten().tolist(), max_abs=float(d.max()))
        self.transient_events.append(ev)

    # ------------------------------------------------------------------ blocks
    def _res_fw(self, name, x, n, H, W, cin, cout):
        a1, c1 = self._gn_fw(name + '.norm1', x, n, H * W, cin, True)
        h, k1 = self._conv_fw(name + '.conv1', a1, n, H, W)
        a2, c2 = self._gn_fw(name + '.norm2', h, n, H * W, cout, True)
        ks = None
        sc = x
        if cin != cout:
            sc, ks = self._conv_fw(name + '.nin_shortcut', x, n, H, W)
        out, k2 = self._conv_fw(name + '.conv2', a2, n, H, W, res=sc)
        return out, (c1, k1, c2, k2, ks)

    def _res_bw(self, ctx, dout):
        c1, k1, c2, k2, ks = ctx
        da2 = self._conv_bw(k2, dout)
        dh = self._gn_bw(c2, da2)
        da1 = self._conv_bw(k1, dh)
        dx = self._gn_bw(c1, da1)
        T.add_(dx, dout if ks is None else self._conv_bw(ks, dout))
        return dx

    def _attn_fw(self, name, x, n, H, W, C):
        """AttnBlock.forward, vqgan_th.py:120-144 (single head, scale C^-0.5)"""
        HW, M = H * W, n * H * W
        hn, cn = self._gn_fw(name + '.norm', x, n, HW, C, False)
        wqkv = torch.cat([self.p(f'{name}.{t}.weight').reshape(C, C) for t in ('q', 'k', 'v')], 0).contiguous()     # [3C][C]
        bqkv = torch.cat([self.p(f'{name}.{t}.bias') for t in ('q', 'k', 'v')], 0).contiguous()
        qkv, _, _ = self._conv_launch(hn, wqkv.view(3 * C, C, 1, 1), bqkv, n, H, W, ops.MODE_GEMM)
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        kp = ops.pack(k, C, HW, 1, sk=1, sn=3 * C, st=0, batch=n, src_bstride=HW * 3 * C)
        S = torch.empty((n, HW, HW), dtype=torch.float32, device=x.device)
        ops.igemm(q, kp, HW, C, HW, S, lda=3 * C, batch=n, stride_x=HW * 3 * C, stride_w=ops.packed_floats(C, HW), stride_out=HW * HW)
        scale = float(int(C) ** (-0.5))
        ops.softmax_rows_(S, n * HW, HW, scale)                                                                      # P
        vp = ops.pack(v, HW, C, 1, sk=3 * C, sn=1, st=0, batch=n, src_bstride=HW * 3 * C)
        a = torch.empty((M, C), dtype=torch.float32, device=x.device)
        ops.igemm(S, vp, HW, HW, C, a, lda=HW, batch=n, stride_x=HW * HW, stride_w=ops.packed_floats(HW, C), stride_out=HW * C)
        out, kpj = self._conv_fw(name + '.proj_out', a, n, H, W, mode=ops.MODE_GEMM, res=x)
        return out, (name, cn, hn, qkv, S, a, kpj, wqkv, n, H, W, C, scale)

    def _attn_bw(self, ctx, dout):
        name, cn, hn, qkv, P, a, kpj, wqkv, n, H, W, C, scale = ctx
        HW, M = H * W, n * H * W
        dev = dout.device
        da = self._conv_bw(kpj, dout)                                                     # [M][C]
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        dqkv = torch.empty((M, 3 * C), dtype=torch.float32, device=dev)
        # dP = da . v^T ; dv = P^T . da
        vtp = ops.pack(v, C, HW, 1, sk=1, sn=3 * C, st=0, batch=n, src_bstride=HW * 3 * C)                # B[c][key] = v[key][c]
        dP = torch.empty((n, HW, HW), dtype=torch.float32, device=dev)
        ops.igemm(da, vtp, HW, C, HW, dP, lda=C, batch=n, stride_x=HW * C, stride_w=ops.packed_floats(C, HW), stride_out=HW * HW)
        Pt = T.transpose(P, HW, HW, batch=n, bs_src=HW * HW)
        dap = ops.pack(da, HW, C, 1, sk=C, sn=1, st=0, batch=n, src_bstride=HW * C)
        ops.igemm(Pt, dap, HW, HW, C, dqkv[:, 2 * C:], lda=HW, ldc=3 * C, batch=n, stride_x=HW * HW, stride_w=ops.packed_floats(HW, C),
                  stride_out=HW * 3 * C)
        T.softmax_rows_bwd_(P, dP, n * HW, HW, scale)                                     # dS (in dP), incl. the C^-0.5 scale
        kq = ops.pack(k, HW, C, 1, sk=3 * C, sn=1, st=0, batch=n, src_bstride=HW * 3 * C)                 # B[key][c] = k[key][c]
        ops.igemm(dP, kq, HW, HW, C, dqkv[:, :C], lda=HW, ldc=3 * C, batch=n, stride_x=HW * HW, stride_w=ops.packed_floats(HW, C),
                  stride_out=HW * 3 * C)                                                  # dq = dS . k
        dSt = T.transpose(dP, HW, HW, batch=n, bs_src=HW * HW)
        qq = ops.pack(q, HW, C, 1, sk=3 * C, sn=1, st=0, batch=n, src_bstride=HW * 3 * C)
        ops.igemm(dSt, qq, HW, HW, C, dqkv[:, C:2 * C], lda=HW, ldc=3 * C, batch=n, stride_x=HW * HW, stride_w=ops.packed_floats(HW, C),
                  stride_out=HW * 3 * C)                                                  # dk = dS^T . q
        # fused q|k|v projection: bias / weight grads of the three 1x1 convs, then d(hn)
        db = torch.zeros(3 * C, dtype=torch.float32, device=dev)
        T.colsum(dqkv, db, M, 3 * C, accumulate=True)
        hnt = T.gather_transpose(hn, 1, 1, M, C, 1, M)                                    # [C][M]
        dw = self._gemm(hnt, dqkv, C, M, 3 * C).t().contiguous()                          # [3C][C]
        for i, t in enumerate(('q', 'k', 'v')):
            T.add_(self.g(f'{name}.{t}.bias'), db[i * C:(i + 1) * C].contiguous())
            T.add_(self.g(f'{name}.{t}.weight').view(C, C), dw[i * C:(i + 1) * C].contiguous())
        dhn = self._gemm(dqkv, wqkv, M, 3 * C, C)
        dx = self._gn_bw(cn, dhn)
        T.add_(dx, dout)
        return dx

    # ------------------------------------------------------------------ plans
    def _plan_fw(self, plan, x, n, H, W, image=None):
        tape = []
        pending = None
        for kind, name, args in plan:
            if kind == 'conv3':
                if args[0] % 32:                                                          # conv_in on the 3-channel image
                    y = ops.conv_in(image, self.p(name + '.weight'), self.p(name + '.bias'), n, H, W, args[1]).view(n * H * W, args[1])
                    tape.append(('conv_in', (name, image, n, H, W, args[1])))
                    x = y
                else:
                    if pending is not None:
                        a, cg = pending
                        pending = None
                    else:
                        a, cg = x, None
                    x, ck = self._conv_fw(name, a, n, H, W)
                    tape.append(('norm_conv', (cg, ck)))
            elif kind == 'res':
                x, c = self._res_fw(name, x, n, H, W, args[0], args[1])
                tape.append(('res', c))
            elif kind == 'attn':
                x, c = self._attn_fw(name, x, n, H, W, args[0])
                tape.append(('attn', c))
            elif kind == 'down':
                x, c = self._conv_fw(name, x, n, H, W, mode=ops.MODE_CONV3_S2PAD)
                H, W = H // 2, W // 2
                tape.append(('conv', c))
            elif kind == 'up':
                x, c = self._conv_fw(name, x, n, H, W, mode=ops.MODE_CONV3_UP2)
                H, W = H * 2, W * 2
                tape.append(('conv', c))
            elif kind == 'norm_swish':
                pending = self._gn_fw(name, x, n, H * W, args[0], True)
        return x, H, W, tape

    def _plan_bw(self, tape, dx):
        for kind, c in reversed(tape):
            if kind == 'res':
                dx = self._res_bw(c, dx)
            elif kind == 'attn':
                dx = self._attn_bw(c, dx)
            elif kind == 'conv':
                dx = self._conv_bw(c, dx)
            elif kind == 'norm_conv':
                cg, ck = c
                dx = self._conv_bw(ck, dx)
                if cg is not None:
                    dx = self._gn_bw(cg, dx)
            elif kind == 'conv_in':
                name, image, n, H, W, cout = c
                P = n * H * W
                T.colsum(dx, self.g(name + '.bias'), P, cout, accumulate=True)
                xt = torch.empty((9, 3, P), dtype=torch.float32, device=dx.device)
                for ky in range(3):
                    for kx in range(3):
                        T.gather_transpose(image, n, H, W, 3, H, W, 1, ky - 1, kx - 1, out=xt[ky * 3 + kx])
                dw = self._gemm(xt.view(27, P), dx, 27, P, cout)                          # [27][cout]
                T.add_(self.g(name + '.weight'), dw.view(3, 3, 3, cout).permute(3, 2, 0, 1).contiguous())
                dx = None
        return dx

    # ------------------------------------------------------------------ the step
    def train_step(self, images, reduce_gradients: bool = True, apply_update: bool = True):
        """images: NCHW float32 in [-1, 1] (the Torch convention of train_codebook_th.py:8-9) or NHWC uint8.  Returns the metrics of
        ``_compute_loss`` (total_loss, rec_loss, quant_loss)."""
        cfg, dev = self.cfg, self.dev
        x = images.to(dev)
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != cfg.in_channels:
            raise TypeError('images must be float32 [N, 3, H, W] in [-1, 1] (train_codebook_th.py:8-9 converts in the data module)')
        n, _, H, W = x.shape
        img = x.permute(0, 2, 3, 1).contiguous()                                          # NHWC float
        self.flat_g.zero_()
        # ---- forward (training mode, vqgan_th.py:349-352) --------------------------------------------------
        h, eh, ew, enc_tape = self._plan_fw(self._enc_plan, None, n, H, W, image=img)
        z, kq = self._conv_fw('quant_conv', h, n, eh, ew, mode=ops.MODE_GEMM)              # [n*eh*ew][D]
        D = cfg.embed_dim
        # lookup against the current codebook, then its EMA update (utils_th.py:32-68); ``quant`` is the straight-through value
        quant, diff, ind = self.quantizer(z.view(n, eh, ew, D).permute(0, 3, 1, 2))
        qrows = quant.permute(0, 2, 3, 1).reshape(n * eh * ew, D)
        d0, kpq = self._conv_fw('post_quant_conv', qrows, n, eh, ew, mode=ops.MODE_GEMM)
        xrec, _, _, dec_tape = self._plan_fw(self._dec_plan, d0, n, eh, ew)
        self.last_indices = ind
        # ---- loss (vqgan_th.py:400-408): mean(|x - xrec| + pw * lpips[n]) + cw * commitment -----------------
        numel = img.numel()
        l1_sum, dxrec = T.l1_loss(img.view(-1), xrec.view(-1), 1.0 / numel)
        rec = l1_sum / numel
        p_mean = torch.zeros((), dtype=torch.float32, device=dev)
        if self.lpips is not None:
            p, dperc = self.lpips.loss_and_grad(img, xrec.view(n, H, W, 3), cfg.perceptual_weight / n)
            p_mean = p.mean()
            rec = rec + cfg.perceptual_weight * p_mean
            T.add_(dxrec, dperc.reshape(-1))
        loss = rec + cfg.codebook_weight * diff
        # ---- backward --------------------------------------------------------------------------------------
        dd0 = self._plan_bw(dec_tape, dxrec.view(n * H * W, cfg.out_ch))
        dq = self._conv_bw(kpq, dd0)                                                      # straight-through: d(quant) -> dz
        zn = z.numel()
        dz = T.axpby(2.0 * cfg.codebook_weight / zn, z, -2.0 * cfg.codebook_weight / zn, qrows)      # d/dz of cw * mean((q - z)^2)
        T.add_(dz, dq)
        dh = self._conv_bw(kq, dz)
        self._plan_bw(enc_tape, dh)
        if reduce_gradients and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.group)          # DDP: mean over replicas
            T.axpby(1.0 / dist.get_world_size(self.group), self.flat_g, out=self.flat_g)
        if apply_update:
            self.apply_gradients()
        return dict(total_loss=loss, rec_loss=rec, quant_loss=diff, p_loss=p_mean)

    def apply_gradients(self):
        """torch.optim.Adam(lr, betas=(0.5, 0.9)), vqgan_th.py:427-429"""
        clip = float(getattr(self.cfg, "gradient_clip_val", 0.0) or 0.0)
        if clip > 0:                                 # pl.Trainer(gradient_clip_val=...), train_codebook_th.py:69: global 2-norm clip
            if getattr(self, '_clip_scratch', None) is None:
                self._clip_scratch = torch.zeros(1, dtype=torch.float32, device=self.flat_g.device)
            T.clip_grad_norm_(self.flat_g, clip, self._clip_scratch)
        self.step_count += 1
        t = self.step_count
        c2 = math.sqrt(1.0 - self.b2 ** t)
        lr_adam = self.lr * c2 / (1.0 - self.b1 ** t)
        # p -= lr/(1-b1^t) * m / (sqrt(v/(1-b2^t)) + eps)  ==  p -= lr_adam * m / (sqrt(v) + eps * sqrt(1-b2^t))
        T.adamw_(self.flat_p, self.flat_g, self.flat_m, self.flat_v, 0.0, lr_adam, self.b1, self.b2, self.eps * c2)
