"""Camera-pose bookkeeping of the evaluators: O(B*S) fp32 math on [B,S,7] tensors
(xyz + quaternion w,x,y,z).  Host-side by design: 7 floats per view,
no kernel.  Works on CPU or GPU tensors alike.

Mirrors viewformer/utils/geometry_tf.py:6-13,44-50,53-68,71-91,
viewformer/evaluate/evaluate_transformer.py:70-94 and viewformer/models/migt.py:123-129,
139-145,150-164.
"""
import torch


def quaternion_multiply(q1, q2):
    w1, x1, y1, z1 = q1.unbind(-1)
    w2, x2, y2, z2 = q2.unbind(-1)
    return torch.stack((-x1 * x2 - y1 * y2 - z1 * z2 + w1 * w2,
                        x1 * w2 + y1 * z2 - z1 * y2 + w1 * x2,
                        -x1 * z2 + y1 * w2 + z1 * x2 + w1 * y2,
                        x1 * y2 - y1 * x2 + z1 * w2 + w1 * z2), -1)


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
