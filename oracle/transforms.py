"""Rotation-representation helpers with pytorch3d.transforms semantics (oracle, torch, any dtype).

pytorch3d==0.7.7 is not vendored under /root/reference (README.md:30); these follow its published
conventions: real-first quaternions (w,x,y,z); ``quaternion_to_matrix`` scales by 2/|q|^2 (no unit
assumption); ``matrix_to_quaternion`` evaluates the four sqrt candidates, picks the
best-conditioned one (largest |component|, denominators floored at 0.1) and returns w >= 0;
``quaternion_invert`` is the conjugate.  Parity UNPINNED by the reference; cross-checked against
scipy.spatial.transform.Rotation in tests/test_oracle_golden.py (test_transforms_against_scipy).
"""
import torch


def quaternion_to_matrix(q: torch.Tensor) -> torch.Tensor:
    w, x, y, z = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    rows = (
        1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
        s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
        s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y),
    )
    return torch.stack(rows, -1).reshape(q.shape[:-1] + (3, 3))


def _sqrt_pos(v: torch.Tensor) -> torch.Tensor:
    out = torch.zeros_like(v)
    pos = v > 0
    out[pos] = torch.sqrt(v[pos])
    return out


def standardize_quaternion(q: torch.Tensor) -> torch.Tensor:
    return torch.where(q[..., 0:1] < 0, -q, q)


def matrix_to_quaternion(m: torch.Tensor) -> torch.Tensor:
    batch = m.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.reshape(batch + (9,)).unbind(-1)
    qa = _sqrt_pos(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22,
                                1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], -1))
    cand = torch.stack([
        torch.stack([qa[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
        torch.stack([m21 - m12, qa[..., 1] ** 2, m10 + m01, m02 + m20], -1),
        torch.stack([m02 - m20, m10 + m01, qa[..., 2] ** 2, m12 + m21], -1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, qa[..., 3] ** 2], -1)], -2)
    floor = torch.tensor(0.1, dtype=qa.dtype, device=qa.device)
    cand = cand / (2.0 * qa[..., None].max(floor))
    pick = torch.nn.functional.one_hot(qa.argmax(-1), num_classes=4) > 0.5
    return standardize_quaternion(cand[pick, :].reshape(batch + (4,)))


def quaternion_raw_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz,
                        aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_invert(q: torch.Tensor) -> torch.Tensor:
    return q * q.new_tensor([1, -1, -1, -1])


# --- the two extra representations mlp_reg.py:72-90 can select (--r rpy / --r 6d) ------------
def _axis_rot(axis: str, a: torch.Tensor) -> torch.Tensor:
    c, s, o, z = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
    flat = {"X": (o, z, z, z, c, -s, z, s, c), "Y": (c, z, s, z, o, z, -s, z, c),
            "Z": (c, -s, z, s, c, z, z, z, o)}[axis]
    return torch.stack(flat, -1).reshape(a.shape + (3, 3))


def euler_angles_to_matrix(e: torch.Tensor, convention: str) -> torch.Tensor:
    mats = [_axis_rot(c, a) for c, a in zip(convention, e.unbind(-1))]
    return mats[0] @ mats[1] @ mats[2]


def matrix_to_euler_angles(m: torch.Tensor, convention: str) -> torch.Tensor:
    if convention != "XYZ":
        raise NotImplementedError("only the XYZ convention is on the reference path")
    # R = Rx(a) Ry(b) Rz(c):  R02 = sin b, R12 = -sin a cos b, R22 = cos a cos b,
    #                         R01 = -cos b sin c, R00 = cos b cos c
    b = torch.asin(m[..., 0, 2])
    a = torch.atan2(-m[..., 1, 2], m[..., 2, 2])
    c = torch.atan2(-m[..., 0, 1], m[..., 0, 0])
    return torch.stack((a, b, c), -1)


def matrix_to_rotation_6d(m: torch.Tensor) -> torch.Tensor:
    return m[..., :2, :].clone().reshape(m.shape[:-2] + (6,))


def rotation_6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), -2)
