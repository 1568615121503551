#!/usr/bin/env python
"""Golden vectors for the image resize of the evaluators' pre-process FROM THE REFERENCE ITSELF (viewformer/data/_common.py:19-61:
``resize`` -> ``resize_th``: uint8 -> /255 -> torch interpolate (nearest when enlarging, bilinear align_corners=False when shrinking)
-> clamp -> *255 -> uint8 truncation).  Run in the build container only:   python tests/golden/make_resize_golden.py
Only data is written: random uint8 inputs and the reference's outputs."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference      # noqa: E402


def main():
    import_reference()
    from viewformer.data._common import resize
    rng = np.random.default_rng(21)
    out = {}
    for i, (n, src, dst, method) in enumerate([(2, 256, 128, None), (2, 200, 128, None), (3, 64, 128, None), (2, 96, 128, None),
                                               (2, 160, 32, None), (1, 48, 128, 'bilinear'), (1, 300, 128, 'nearest'),
                                               (2, 128, 128, None)]):
        img = rng.integers(0, 256, size=(n, src, src, 3), dtype=np.uint8)
        if i % 2 == 0 and src % 8 == 0:                      # smooth content too: flat regions are where truncation after *255 bites
            img = (img.astype(np.float32).reshape(n, src // 8, 8, src // 8, 8, 3).mean((2, 4), keepdims=True)
                   .repeat(8, 2).repeat(8, 4).reshape(n, src, src, 3)).astype(np.uint8)
        out[f'in{i}'] = img
        out[f'out{i}'] = resize(img, dst, method)
        out[f'meta{i}'] = np.array([dst, -1 if method is None else ['nearest', 'bilinear'].index(method)])
    np.savez_compressed(os.path.join(HERE, 'resize.npz'), **out)
    print({k: v.shape for k, v in out.items() if k.startswith('out')})


if __name__ == '__main__':
    main()
