"""CPU: the evaluators' camera bookkeeping (viewformer_amd/geometry.py).  The Hamilton product is evaluated as a gathered
[...,4,4] term table (7 element-wise launches instead of 32 per product); its results must be the literal formula's
(viewformer/utils/geometry_tf.py:6-13) bit for bit, because the relative cameras feed the transformer's pose embedding and the tokens are
compared bit-exactly."""
import numpy as np
import torch

from viewformer_amd import geometry


def _literal_product(q1, q2):
    w1, x1, y1, z1 = q1.unbind(-1)
    w2, x2, y2, z2 = q2.unbind(-1)
    return torch.stack((-x1 * x2 - y1 * y2 - z1 * z2 + w1 * w2,
                        x1 * w2 + y1 * z2 - z1 * y2 + w1 * x2,
                        -x1 * z2 + y1 * w2 + z1 * x2 + w1 * y2,
                        x1 * y2 - y1 * x2 + z1 * w2 + w1 * z2), -1)


def _bits(t):
    return t.contiguous().view(torch.int32 if t.dtype == torch.float32 else torch.int64)


def test_quaternion_product_is_the_literal_formula_bit_for_bit():
    g = torch.Generator().manual_seed(5)
    for dtype in (torch.float32, torch.float64):
        for shape in ((4,), (3, 4), (128, 7, 4), (2, 5, 1, 4)):
            a = torch.randn(*shape, generator=g, dtype=dtype) * 3
            b = torch.randn(*shape, generator=g, dtype=dtype) * 0.01
            assert torch.equal(_bits(geometry.quaternion_multiply(a, b)), _bits(_literal_product(a, b)))
    # broadcasting operands (the frame changes expand the first view's rotation), zeros and signed zeros (the rotation embeds a point with w = 0)
    a = torch.randn(6, 1, 4, generator=g)
    b = torch.randn(6, 7, 4, generator=g)
    b[..., 0] = 0.0
    b[0, 0, 1:] = -0.0
    assert torch.equal(_bits(geometry.quaternion_multiply(a.expand_as(b), b)), _bits(_literal_product(a.expand_as(b), b)))
    assert torch.equal(_bits(geometry.quaternion_multiply(a, b)), _bits(_literal_product(a.expand_as(b), b)))
    # non-contiguous views (cameras[..., 3:] is a slice of the [B,S,7] tensor)
    cams = torch.randn(9, 7, 7, generator=g)
    q = cams[..., 3:]
    assert torch.equal(_bits(geometry.quaternion_multiply(q, q.flip(1))), _bits(_literal_product(q, q.flip(1))))


def test_frame_change_round_trip_and_oracle_agreement():
    from oracle import migt_oracle as O
    g = torch.Generator().manual_seed(11)
    cams = torch.randn(16, 7, 7, generator=g)
    cams[..., 3:] = geometry.quaternion_normalize(cams[..., 3:])
    rel, tr = geometry.to_relative_cameras(cams)
    back = geometry.from_relative_cameras(rel, tr)
    assert torch.allclose(back, cams, atol=2e-6)
    # first view of every scene: identity pose
    assert torch.allclose(rel[:, 0, :3], torch.zeros(16, 3), atol=1e-6) and torch.allclose(rel[:, 0, 3].abs(), torch.ones(16), atol=1e-6)
    o_rel, o_tr = O.to_relative_cameras(cams)
    assert torch.equal(_bits(torch.as_tensor(np.asarray(o_rel), dtype=torch.float32)), _bits(rel))
    assert torch.equal(_bits(torch.as_tensor(np.asarray(o_tr), dtype=torch.float32)), _bits(tr))
