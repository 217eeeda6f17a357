"""Dual-quaternion helpers, oracle restatement of reference PointCloud/dq_func.py:4-257.

Layout: (...,8) = [real (w,x,y,z) | dual (w,x,y,z)].  Quirks that are preserved on purpose:
``dualquat_to_quat_trans`` returns real (x) dual as its "q" (dq_func.py:144);
``dualquat_to_rot_trans`` uses the conjugate (not the inverse) of the real part (dq_func.py:167)
and a 2/|q|^2-scaled rotation, so non-unit real parts are tolerated, not renormalised.
"""
import torch

from . import transforms as T


def transform_from_rot_trans(R, t):
    out = torch.zeros(*R.shape[:-2], 4, 4, dtype=R.dtype, device=R.device)
    out[..., :3, :3], out[..., :3, 3], out[..., 3, 3] = R, t, 1.0
    return out


def quaternion_conjugate(q):
    return torch.cat([q[..., :1], -q[..., 1:]], -1)


def quat_trans_to_dualquat(q, t):
    pure = torch.cat([torch.zeros_like(q[..., :1]), t], -1)
    return torch.cat([q, 0.5 * T.quaternion_raw_multiply(pure, q)], -1)


def rot_trans_to_dualquat(R, t):
    q = T.matrix_to_quaternion(R)
    n = torch.linalg.norm(q, dim=-1, keepdim=True)
    return quat_trans_to_dualquat(q / torch.clamp_min(n, torch.finfo(R.dtype).eps), t)


def transform_to_dualquat(M):
    return rot_trans_to_dualquat(M[..., :3, :3], M[..., :3, 3])


def _trans_of(dq):
    re, du = dq[..., :4], dq[..., 4:]
    return (2 * T.quaternion_raw_multiply(du, T.quaternion_invert(re)))[..., 1:]


def dualquat_to_quat_trans(dq):
    return T.quaternion_raw_multiply(dq[..., :4], dq[..., 4:]), _trans_of(dq)


def dualquat_to_rot_trans(dq):
    return T.quaternion_to_matrix(dq[..., :4]), _trans_of(dq)


def dualquat_to_transform(dq):
    return transform_from_rot_trans(*dualquat_to_rot_trans(dq))


def dualquat_multiply(a, b):
    ar, ad, br, bd = a[..., :4], a[..., 4:], b[..., :4], b[..., 4:]
    mul = T.quaternion_raw_multiply
    return torch.cat([mul(ar, br), mul(ar, bd) + mul(ad, br)], -1)


def dualquat_invert(dq):
    eps = torch.finfo(dq.dtype).eps
    re, du = dq[..., :4], dq[..., 4:]
    n2 = torch.clamp_min(torch.linalg.norm(re, dim=-1, keepdim=True) ** 2, eps)
    rc = quaternion_conjugate(re)
    dot = (re * du).sum(-1, keepdim=True) / n2 ** 2
    return torch.cat([rc / n2, quaternion_conjugate(du) / n2 - 2 * rc * dot], -1)


def point_to_dualquat(p):
    out = torch.zeros(*p.shape[:-1], 8, dtype=p.dtype, device=p.device)
    out[..., 0] = 1.0
    out[..., 5:] = p
    return out
