"""Camera-pose bookkeeping of the evaluators: O(B*S) fp32 math on [B,S,7] tensors
(xyz + quaternion w,x,y,z).  Host-side by design: 7 floats per view,
no kernel.  Works on CPU or GPU tensors alike.

Mirrors viewformer/utils/geometry_tf.py:6-13,44-50,53-68,71-91,
viewformer/evaluate/evaluate_transformer.py:70-94 and viewformer/models/migt.py:123-129,
139-145,150-164.
"""
import torch


# Hamilton product, term table of geometry_tf.py:6-13: output component o = sum over t of  sign[o][t] * q1[(t+1)%4] * q2[_Q2_IDX[o][t]],
# summed left to right.  One gather + one multiply per operand and three adds (7 element-wise launches) instead of one launch per scalar
# operation of the literal formula (32): the evaluator's frame changes were ~180 launches of 4-5 us per inference step.  The results are the
# literal formula's bit for bit — a sign flip is exact, and a - b == a + (-b) in IEEE arithmetic (tests/test_geometry.py).
_Q1_IDX = (1, 2, 3, 0)
_Q2_IDX = ((1, 2, 3, 0), (0, 3, 2, 1), (3, 0, 1, 2), (2, 1, 0, 3))
_Q_SIGN = ((-1., -1., -1., 1.), (1., 1., -1., 1.), (-1., 1., 1., 1.), (1., -1., 1., 1.))
_q_tables = {}


def _quaternion_tables(device, dtype):
    key = (str(device), dtype)
    if key not in _q_tables:
        _q_tables[key] = (torch.tensor(_Q1_IDX, device=device), torch.tensor(_Q2_IDX, device=device).reshape(-1),
                          torch.tensor(_Q_SIGN, device=device, dtype=dtype))
    return _q_tables[key]


def quaternion_multiply(q1, q2):
    q1, q2 = torch.broadcast_tensors(q1, q2)
    i1, i2, sign = _quaternion_tables(q1.device, q1.dtype)
    a = q1.index_select(-1, i1).unsqueeze(-2) * sign                               # [...,4 out,4 terms]
    p = a * q2.index_select(-1, i2).unflatten(-1, (4, 4))
    return ((p[..., 0] + p[..., 1]) + p[..., 2]) + p[..., 3]


def quaternion_normalize(x, epsilon: float = 1e-12):
    """tf.linalg.l2_normalize semantics: x * rsqrt(max(sum x^2, eps))"""
    return x * torch.rsqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=epsilon))


def quaternion_remove_sign(x):
    return x * (2 * (x[..., :1] >= 0).to(x.dtype) - 1)


def quaternion_conjugate(q):
    return torch.cat((q[..., :1], -q[..., 1:]), -1)


def quaternion_rotate(point, q):
    p = torch.cat([torch.zeros_like(point[..., :1]), point], -1)
    return quaternion_multiply(quaternion_multiply(q, p), quaternion_conjugate(q))[..., 1:]


def make_quaternion(axis, angle):
    """geometry_tf.py:16-21: (cos(a/2), sin(a/2) * axis); angle [...], axis [3] -> [...,4]"""
    angle = torch.as_tensor(angle)
    axis = torch.as_tensor(axis, dtype=angle.dtype, device=angle.device)
    return torch.cat([torch.cos(angle / 2)[..., None], torch.sin(angle / 2)[..., None] * axis], -1)


def make_quaternion_y(angle):
    return make_quaternion([0.0, 1.0, 0.0], angle)              # geometry_tf.py:24-27


def make_quaternion_x(angle):
    return make_quaternion([1.0, 0.0, 0.0], angle)              # geometry_tf.py:30-33


def to_relative_cameras(cameras):
    """evaluate_transformer.py:70-78 -> (relative cameras, transform of the first view)"""
    xyz, quat = cameras[..., :3], cameras[..., 3:]
    t_xyz, t_quat = xyz[..., :1, :], quat[..., :1, :]
    rinv = quaternion_conjugate(t_quat).expand_as(quat)
    return (torch.cat((quaternion_rotate(xyz - t_xyz, rinv), quaternion_multiply(rinv, quat)), -1),
            torch.cat((t_xyz, t_quat), -1))


def from_relative_cameras(cameras, transform):
    """evaluate_transformer.py:81-87"""
    t_xyz, t_quat = transform[..., :3], transform[..., 3:]
    xyz, quat = cameras[..., :3], cameras[..., 3:]
    tq = t_quat.expand_as(quat)
    return torch.cat((quaternion_rotate(xyz, tq) + t_xyz, quaternion_multiply(tq, quat)), -1)


def normalize_cameras(cameras):
    """evaluate_transformer.py:90-94"""
    return torch.cat((cameras[..., :3], quaternion_remove_sign(quaternion_normalize(cameras[..., 3:]))), -1)


def reduce_cameras(x, axis=-2):
    """QuaternionPoseRepresentation.reduce + quaternion_reduce_mean, migt.py:123-129,150-154"""
    q = quaternion_remove_sign(quaternion_normalize(x[..., 3:])).mean(axis)
    return torch.cat((x[..., :3].mean(axis), quaternion_remove_sign(quaternion_normalize(q))), -1)


def pose_model_input(poses, position_multiplier: float):
    """get_model_input, migt.py:139-145 (random multiplier == 1 at inference)"""
    return torch.cat([poses[..., :3] * position_multiplier, poses[..., 3:]], -1)


def pose_head_postprocess(raw, position_multiplier: float):
    """QuaternionPoseRepresentation.call output branch, migt.py:159-164"""
    q = quaternion_remove_sign(quaternion_normalize(raw[..., 3:]))
    return torch.cat([raw[..., :3] / position_multiplier, q], -1)
