"""L1 Chamfer distance with pytorch3d.loss.chamfer_distance(x, y, norm=1) semantics (oracle).

Reference call site: PointCloud/mlp_reg.py:96 (batch of 1, defaults point_reduction="mean",
batch_reduction="mean", bidirectional).  pytorch3d is not vendored; see creg_oracle.c header.
"""
import ctypes

import numpy as np
import torch

from ._clib import lib


def nn_l1(x: np.ndarray, y: np.ndarray):
    """K=1 L1 nearest neighbour of every row of x (nx,3) among y (ny,3): (dist f32, idx i64)."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    d = np.empty(len(x), np.float32)
    i = np.empty(len(x), np.int64)
    lib().oracle_nn_l1_f32(x.ctypes.data, len(x), y.ctypes.data, len(y), d.ctypes.data, i.ctypes.data)
    return d, i


def nn_l1_bwd(x, y, ix, iy, gx: float, gy: float):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    ix = np.ascontiguousarray(ix, np.int64)
    iy = np.ascontiguousarray(iy, np.int64)
    px = np.empty_like(x)
    py = np.empty_like(x)
    lib().oracle_nn_l1_bwd_f32(x.ctypes.data, len(x), y.ctypes.data, len(y), ix.ctypes.data,
                               iy.ctypes.data, ctypes.c_float(gx), ctypes.c_float(gy),
                               px.ctypes.data, py.ctypes.data)
    return px + py


class _ChamferL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        xn, yn = x.detach().numpy(), y.detach().numpy()
        dx, ix = nn_l1(xn, yn)
        dy, iy = nn_l1(yn, xn)
        ctx.save_for_backward(x.detach(), y.detach())
        ctx.ix, ctx.iy = ix, iy
        cham_x = torch.from_numpy(dx).sum() / len(xn)
        cham_y = torch.from_numpy(dy).sum() / len(yn)
        return cham_x + cham_y

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        gx = float(g) * float(np.float32(1.0) / np.float32(len(x)))
        gy = float(g) * float(np.float32(1.0) / np.float32(len(y)))
        grad = nn_l1_bwd(x.numpy(), y.numpy(), ctx.ix, ctx.iy, gx, gy)
        return torch.from_numpy(grad), None


def chamfer_distance(x: torch.Tensor, y: torch.Tensor, norm: int = 1):
    """x (1,P1,3), y (1,P2,3) fp32 CPU -> (loss, None); gradient flows to x only."""
    if norm != 1 or x.shape[0] != 1 or y.shape[0] != 1:
        raise NotImplementedError("oracle covers the reference call: batch 1, norm=1")
    return _ChamferL1.apply(x[0].contiguous(), y[0].contiguous()), None


def chamfer_l1_dense(x: torch.Tensor, y: torch.Tensor):
    """Independent cross-check: same quantity through torch.cdist(p=1) (different op order)."""
    d = torch.cdist(x, y, p=1)
    dx, ix = d.min(1)
    dy, iy = d.min(0)
    return dx.mean() + dy.mean(), ix, iy
