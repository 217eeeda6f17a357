"""Euler-angle and 6-D rotation maps with pytorch3d.transforms conventions, as torch ops on the GPU.

Only the optional `--r rpy` / `--r 6d` modes of the reference use them (mlp_reg.py:72-76, 86-90);
they are K-row (<= 160) conversions inside the autograd graph of the compatibility train loop, so
they stay in PyTorch-ROCm.  The default `--r q` and `--r dq` modes never touch this module: their
conversions are HIP kernels inside the fused train plan.
"""
import torch


def _axis(axis, a):
    c, s, o, z = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
    flat = {"X": (o, z, z, z, c, -s, z, s, c), "Y": (c, z, s, z, o, z, -s, z, c), "Z": (c, -s, z, s, c, z, z, z, o)}[axis]
    return torch.stack(flat, -1).reshape(a.shape + (3, 3))


def euler_angles_to_matrix(e, convention="XYZ"):
    m = [_axis(c, a) for c, a in zip(convention, e.unbind(-1))]
    return m[0] @ m[1] @ m[2]


def matrix_to_euler_angles(m, convention="XYZ"):
    if convention != "XYZ":
        raise NotImplementedError("only XYZ is used on the registration path")
    return torch.stack((torch.atan2(-m[..., 1, 2], m[..., 2, 2]), torch.asin(m[..., 0, 2]),
                        torch.atan2(-m[..., 0, 1], m[..., 0, 0])), -1)


def matrix_to_rotation_6d(m):
    return m[..., :2, :].clone().reshape(m.shape[:-2] + (6,))


def rotation_6d_to_matrix(d6):
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), -2)
