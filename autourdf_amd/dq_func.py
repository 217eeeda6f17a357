"""Dual-quaternion helpers on the MI355X -- drop-in for reference PointCloud/dq_func.py (same 11
function names and argument meaning, dq_func.py:4,29,47,72,100,126,148,170,188,213,238).

Tensors are (...,8) = [real (w,x,y,z) | dual]; every function takes fp32 CUDA tensors and runs a
HIP row kernel from libcreg.so (K5).  Like the reference's plain-torch functions, all of them are
transparent to autograd: ``dualquat_to_transform`` (inside the reference's graph at mlp_reg.py:83-84)
has its adjoint kernel; for the others the forward is the kernel and the backward differentiates the
same formula in PyTorch-ROCm tensor ops on the device (``_dq_autograd``).
"""
import torch

from . import _dq_autograd as _ag
from . import ops


def _rows(t: torch.Tensor, width):
    if not t.is_cuda:
        raise RuntimeError("autourdf_amd.dq_func runs on the GPU only (no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError("autourdf_amd.dq_func kernels are fp32, like the reference's use of them")
    lead = t.shape[:-len(width)] if isinstance(width, tuple) else t.shape[:-1]
    w = width if isinstance(width, tuple) else (width,)
    return t.reshape((-1,) + w).contiguous(), lead


def transform_from_rot_trans(R: torch.Tensor, t: torch.Tensor):
    assert R.shape[-2:] == (3, 3) and t.shape[-1] == 3
    T = torch.zeros(*R.shape[:-2], 4, 4, device=R.device, dtype=R.dtype)
    T[..., :3, :3], T[..., :3, 3], T[..., 3, 3] = R, t, 1.0
    return T


def quaternion_conjugate(q: torch.Tensor) -> torch.Tensor:
    assert q.shape[-1] == 4
    return q * q.new_tensor([1.0, -1.0, -1.0, -1.0])


def quat_trans_to_dualquat(q: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    assert q.shape[-1] == 4 and t.shape[-1] == 3
    qr, lead = _rows(q, 4)
    tr, _ = _rows(t, 3)
    return _ag.with_torch_backward(ops.quat_trans_to_dq, _ag.quat_trans_to_dualquat, qr, tr).reshape(lead + (8,))


def rot_trans_to_dualquat(R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    assert R.shape[-2:] == (3, 3) and t.shape[-1] == 3
    return transform_to_dualquat(transform_from_rot_trans(R, t))


def transform_to_dualquat(T: torch.Tensor) -> torch.Tensor:
    assert T.shape[-2:] == (4, 4)
    Tr, lead = _rows(T, (4, 4))
    return _ag.with_torch_backward(ops.se3_to_dq, _ag.transform_to_dualquat, Tr).reshape(lead + (8,))


def dualquat_to_quat_trans(dq: torch.Tensor):
    assert dq.shape[-1] == 8
    d, lead = _rows(dq, 8)
    q, t = _ag.with_torch_backward(ops.dq_to_quat_trans, _ag.dualquat_to_quat_trans, d)
    return q.reshape(lead + (4,)), t.reshape(lead + (3,))


class _DqToSe3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d):
        ctx.save_for_backward(d)
        return ops.dq_to_se3(d)

    @staticmethod
    def backward(ctx, gM):
        (d,) = ctx.saved_tensors
        return ops.dq_to_se3_bwd(d, gM.contiguous())


def dualquat_to_transform(dq: torch.Tensor) -> torch.Tensor:
    assert dq.shape[-1] == 8
    d, lead = _rows(dq, 8)
    return _DqToSe3.apply(d).reshape(lead + (4, 4))


def dualquat_to_rot_trans(dq: torch.Tensor):
    T = dualquat_to_transform(dq)
    return T[..., :3, :3], T[..., :3, 3]


def dualquat_multiply(dq1: torch.Tensor, dq2: torch.Tensor) -> torch.Tensor:
    assert dq1.shape[-1] == 8 and dq2.shape[-1] == 8
    a, lead = _rows(dq1, 8)
    b, _ = _rows(dq2, 8)
    return _ag.with_torch_backward(ops.dq_multiply, _ag.dualquat_multiply, a, b).reshape(lead + (8,))


def dualquat_invert(dq: torch.Tensor) -> torch.Tensor:
    assert dq.shape[-1] == 8
    d, lead = _rows(dq, 8)
    return _ag.with_torch_backward(ops.dq_invert, _ag.dualquat_invert, d).reshape(lead + (8,))


def point_to_dualquat(p: torch.Tensor) -> torch.Tensor:
    assert p.shape[-1] == 3
    out = torch.zeros(*p.shape[:-1], 8, device=p.device, dtype=p.dtype)
    out[..., 0] = 1.0
    out[..., 5:] = p
    return out
