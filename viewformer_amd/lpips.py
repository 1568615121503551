"""LPIPS (VGG) perceptual distance on the GPU — what ``lpips.LPIPS(net='vgg')`` computes where the reference calls it: the codebook
training loss (viewformer/models/vqgan_th.py:337-339,402-404; TF twin vqgan.py:270,324-326) and, forward only, the evaluators'
``LPIPSMetric('vgg')`` (evaluate/evaluate_transformer.py:34, evaluate_codebook.py:27).

``lpips`` (PyPI, v0.1.x; requirements of the reference) is a third-party package that is neither in the reference tree nor in this
image, and its weights (torchvision VGG-16 + the learned ``lin`` layers) are downloads.  This module restates the published algorithm:
  ScalingLayer -> VGG-16 features tapped after relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 -> per-pixel channel normalisation
  x / (||x|| + 1e-10) -> squared difference -> 1x1 ``lin`` weights -> spatial mean -> sum over the five taps,
on weights the caller supplies (``load_lpips_weights`` reads the two upstream ``.pth`` files); ``make_lpips_weights`` builds random
weights of the same shapes for tests and benchmarks.  Parity is pinned against oracle/lpips_oracle.py (same restatement in fp64), NOT
against the package: "parity unpinned" until a run with the real package's outputs is recorded.

All arithmetic is in libvf_hip.so: the 13 convolutions on the library's convolution kernels (fp32-equivalent split-bf16 where the
shape allows, fp32 MFMA otherwise), the rest in csrc/lpips.hip.  The weights are frozen (vqgan_th.py:338-339), so the backward pass
only carries dX, to the reconstruction."""
import numpy as np
import torch

from . import ops
from . import train_ops as T

# (slice, [(index in torchvision's vgg16.features, Cin, Cout), ...]); a 2x2 max-pool precedes every slice but the first
VGG_SLICES = [
    (1, [(0, 3, 64), (2, 64, 64)]),
    (2, [(5, 64, 128), (7, 128, 128)]),
    (3, [(10, 128, 256), (12, 256, 256), (14, 256, 256)]),
    (4, [(17, 256, 512), (19, 512, 512), (21, 512, 512)]),
    (5, [(24, 512, 512), (26, 512, 512), (28, 512, 512)]),
]
SHIFT = (-0.030, -0.088, -0.188)       # lpips ScalingLayer
SCALE = (0.458, 0.448, 0.450)


def lpips_keys():
    keys = []
    for s, convs in VGG_SLICES:
        for idx, _, _ in convs:
            keys += [f'net.slice{s}.{idx}.weight', f'net.slice{s}.{idx}.bias']
    return keys + [f'lin{k}.model.1.weight' for k in range(5)]


def make_lpips_weights(seed: int = 0):
    """random weights with the shapes / key names of ``lpips.LPIPS(net='vgg').state_dict()`` (He-normal convolutions, small biases,
    non-negative lin weights as the package clamps them)"""
    g = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for s, convs in VGG_SLICES:
        for idx, cin, cout in convs:
            sd[f'net.slice{s}.{idx}.weight'] = (g.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
            sd[f'net.slice{s}.{idx}.bias'] = (g.standard_normal(cout) * 0.05).astype(np.float32)
    for k, c in enumerate((64, 128, 256, 512, 512)):
        sd[f'lin{k}.model.1.weight'] = (g.uniform(0.0, 2.0 / c, size=(1, c, 1, 1))).astype(np.float32)
    return sd


def load_lpips_weights(vgg16_pth: str, lpips_vgg_pth: str):
    """torchvision ``vgg16-397923af.pth`` (keys ``features.<i>.weight``) + the package's ``weights/v0.1/vgg.pth`` (keys
    ``lin<k>.model.1.weight``) -> the state dict this module takes"""
    from .checkpoint import read_torch_checkpoint
    feats, lins = read_torch_checkpoint(vgg16_pth), read_torch_checkpoint(lpips_vgg_pth)
    sd = {}
    for s, convs in VGG_SLICES:
        for idx, _, _ in convs:
            for t in ('weight', 'bias'):
                sd[f'net.slice{s}.{idx}.{t}'] = np.asarray(feats[f'features.{idx}.{t}'], dtype=np.float32)
    for k in range(5):
        sd[f'lin{k}.model.1.weight'] = np.asarray(lins[f'lin{k}.model.1.weight'], dtype=np.float32)
    return sd


class _Conv:
    __slots__ = ('cin', 'cout', 'w', 'b', 'fw', 'bw')


class LPIPS:
    def __init__(self, state_dict, device):
        missing = [k for k in lpips_keys() if k not in state_dict]
        if missing:
            raise RuntimeError(f'Missing keys: {missing}')
        self.dev = torch.device(device)
        if self.dev.type != 'cuda':
            raise RuntimeError('LPIPS runs on the GPU only (no CPU fallback)')
        self.slices = []
        for s, convs in VGG_SLICES:
            cs = []
            for idx, cin, cout in convs:
                c = _Conv()
                c.cin, c.cout = cin, cout
                c.w = torch.as_tensor(np.ascontiguousarray(state_dict[f'net.slice{s}.{idx}.weight']), dtype=torch.float32).to(self.dev)
                c.b = torch.as_tensor(np.ascontiguousarray(state_dict[f'net.slice{s}.{idx}.bias']), dtype=torch.float32).to(self.dev)
                if tuple(c.w.shape) != (cout, cin, 3, 3):
                    raise RuntimeError(f'net.slice{s}.{idx}.weight: expected {(cout, cin, 3, 3)}, got {tuple(c.w.shape)}')
                c.fw, c.bw = {}, {}           # packings by kernel family, built on first use
                cs.append(c)
            self.slices.append(cs)
        self.lin = [torch.as_tensor(np.ascontiguousarray(state_dict[f'lin{k}.model.1.weight']), dtype=torch.float32).reshape(-1)
                    .contiguous().to(self.dev) for k in range(5)]

    # ------------------------------------------------------------------ convolutions (frozen weights: packed once per kernel family)
    @staticmethod
    def _conv(x, w, bias, cache, n, H, W, activations=False):
        """``activations``: forward features (x3h applies); gradients stay on x6 (no range condition)"""
        cout, cin = w.shape[0], w.shape[1]
        if ops.conv3_small_cout_supported(ops.MODE_CONV3_S1, cin, cout, H, W):
            return ops.conv3_small_cout(x, w, bias, n, H, W, cin, cout)
        out = torch.empty((n * H * W, cout), dtype=torch.float32, device=x.device)
        x6 = ops.conv3_x6_supported(ops.MODE_CONV3_S1, cin, cout, H, W)
        x3h = x6 and activations
        key = 'x3h' if x3h else 'x6' if x6 else 'f32'
        if key not in cache:
            cache[key] = ops.pack_conv3_x3h(w) if x3h else ops.pack_conv3_x6(w) if x6 else ops.pack_conv_oihw(w)
        ops.igemm(x, cache[key], n * H * W, cin, cout, out, bias=bias, mode=ops.MODE_CONV3_S1, Hin=H, Win=W, Hout=H, Wout=W,
                  x6=x6 and not x3h, x3h=x3h)
        return out

    def _features(self, img, n, H, W):
        """img NHWC float [n,H,W,3] in [-1,1] -> (five taps [(rows, h, w, C)], per-conv outputs); every conv output is post-ReLU"""
        if H % 16 or W % 16:
            raise ValueError('LPIPS (VGG): image sides must be multiples of 16')
        x = T.lpips_scaling(img.reshape(-1, 3), SHIFT, SCALE)
        taps, outs = [], []
        h, w = H, W
        for si, convs in enumerate(self.slices):
            if si > 0:
                h, w = h // 2, w // 2
                x = T.maxpool2(x, n, h, w, convs[0].cin)
            so = []
            for c in convs:
                if c.cin == 3:
                    x = ops.conv_in(x.view(n, h, w, 3), c.w, c.b, n, h, w, c.cout).view(n * h * w, c.cout)
                else:
                    x = self._conv(x, c.w, c.b, c.fw, n, h, w, activations=True)
                T.relu_(x)
                so.append(x)
            outs.append(so)
            taps.append((x, h, w, convs[-1].cout))
        return taps, outs

    def _distances(self, taps, n):
        p = torch.zeros(n, dtype=torch.float32, device=self.dev)
        for k, (f, h, w, c) in enumerate(taps):
            HW = h * w
            sums = T.lpips_head(f[:n * HW], f[n * HW:], self.lin[k], n, HW, c)
            T.axpby(1.0, p, 1.0 / HW, sums, out=p)
        return p

    def __call__(self, x, y):
        """x, y: NHWC float32 [N,H,W,3] in [-1,1] on the GPU -> distances [N] (LPIPS.forward with normalize=False)"""
        n, H, W, _ = x.shape
        taps, _ = self._features(torch.cat([x, y], 0).contiguous(), 2 * n, H, W)
        return self._distances(taps, n)

    def loss_and_grad(self, x, xrec, grad_weight):
        """-> (distances [N], d/d xrec of grad_weight * sum_n distance[n]), NHWC"""
        n, H, W, _ = x.shape
        taps, outs = self._features(torch.cat([x, xrec], 0).contiguous(), 2 * n, H, W)
        p = self._distances(taps, n)
        d = None
        for k in reversed(range(5)):
            f, h, w, c = taps[k]
            HW = h * w
            f0, f1 = f[:n * HW], f[n * HW:]
            if d is None:
                d = torch.empty_like(f1)
            T.lpips_head_bwd(f0, f1, self.lin[k], d, n * HW, c, grad_weight / HW, accumulate=(k != 4))
            for ci in reversed(range(len(self.slices[k]))):
                cv = self.slices[k][ci]
                T.relu_bwd_(d, outs[k][ci][n * HW:])
                if 'rot' not in cv.bw:
                    cv.bw['rot'] = cv.w.flip(2, 3).permute(1, 0, 2, 3).contiguous()             # dX = conv with this weight
                    cv.bw['cache'] = {}
                d = self._conv(d, cv.bw['rot'], None, cv.bw['cache'], n, h, w)
            if k > 0:
                pf, ph, pw_, pc = taps[k - 1]
                d = T.maxpool2_bwd(pf[n * ph * pw_:], d, n, h, w, pc)
        dimg = T.lpips_scaling(d, SHIFT, SCALE, backward=True)
        return p, dimg.view(n, H, W, 3)
