"""ORACLE (test infrastructure, NOT product code) — CPU restatement of ``lpips.LPIPS(net='vgg')`` as the reference calls it
(viewformer/models/vqgan_th.py:337-339,402-404: frozen, inputs in [-1, 1], default ``normalize=False``).

PARITY UNPINNED: ``lpips`` (PyPI v0.1.x) is a third-party dependency that is absent from /root/reference and from this image, and
its weights are downloads; no output of the real package could be recorded.  This restates the package's published algorithm
(lpips/lpips.py: ScalingLayer, vgg16 slices at torchvision ``features`` indices 0-3 / 4-8 / 9-15 / 16-22 / 23-29, normalize_tensor
with eps 1e-10, NetLinLayer = 1x1 conv without bias, spatial_average, sum over layers).  Only ``tests/`` may import this module."""
import torch
import torch.nn.functional as F

SLICES = [(1, [0, 2]), (2, [5, 7]), (3, [10, 12, 14]), (4, [17, 19, 21]), (5, [24, 26, 28])]
SHIFT = (-0.030, -0.088, -0.188)
SCALE = (0.458, 0.448, 0.450)


def features(sd, x, dtype=torch.float64):
    """x NCHW in [-1, 1] -> [relu1_2, relu2_2, relu3_3, relu4_3, relu5_3]"""
    shift = torch.tensor(SHIFT, dtype=dtype).view(1, 3, 1, 1)
    scale = torch.tensor(SCALE, dtype=dtype).view(1, 3, 1, 1)
    h = (x.to(dtype) - shift) / scale
    taps = []
    for s, idxs in SLICES:
        if s > 1:
            h = F.max_pool2d(h, 2, 2)
        for i in idxs:
            h = F.relu(F.conv2d(h, torch.as_tensor(sd[f'net.slice{s}.{i}.weight']).to(dtype),
                                torch.as_tensor(sd[f'net.slice{s}.{i}.bias']).to(dtype), padding=1))
        taps.append(h)
    return taps


def normalize_tensor(f, eps=1e-10):
    return f / (torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True)) + eps)


def distance(sd, x0, x1, dtype=torch.float64):
    """-> [N] (the package returns [N,1,1,1])"""
    f0, f1 = features(sd, x0, dtype), features(sd, x1, dtype)
    val = 0
    for k in range(5):
        d = (normalize_tensor(f0[k]) - normalize_tensor(f1[k])) ** 2
        w = torch.as_tensor(sd[f'lin{k}.model.1.weight']).to(dtype).view(1, -1, 1, 1)
        val = val + (d * w).sum(1, keepdim=True).mean((2, 3), keepdim=True)
    return val.view(-1)
