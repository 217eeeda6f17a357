"""The two registration primitives ``Sim/evaluation.py`` of the reference runs per predicted / ground
truth cloud pair, on the kernels of the hot path: the ICP filter (:358-362,
point-to-point, threshold 0.01, identity start, max 20000 iterations) and ``torch_chamfer_distance``
(:69-81, L1 Chamfer in float32).  URDF / PyBullet joint evaluation is out of scope.  No CPU fallback."""
import numpy as np
import torch

from . import _lib, ops
from .cluster_icp import PointCloud


def _points(p):
    return np.asarray(p.points if hasattr(p, "points") else p, dtype=np.float64).reshape(-1, 3)


def torch_chamfer_distance(p1, p2):
    """p1, p2: point clouds (objects with ``.points`` or arrays).  L1 Chamfer distance in float32 (K1)."""
    dev = _lib.device()
    a = torch.as_tensor(_points(p1), dtype=torch.float32, device=dev)
    b = torch.as_tensor(_points(p2), dtype=torch.float32, device=dev)
    return ops.chamfer_distance(a[None], b[None], norm=1)[0].item()


def icp_filter(pred_pcd, gt_pcd, threshold=0.01, max_iteration=20000):
    """registration_icp(pred, gt, threshold, I, point-to-point) and pred moved by the result.
    Returns (transformation (4,4) float64, moved PointCloud).  Clouds above 1024 source points / 65536 targets run in K4's
    many-workgroup regime (round 5: point-to-point mode there too -- 64 sources per workgroup, cell grid over the target cloud),
    smaller ones as one workgroup."""
    dev = _lib.device()
    src = torch.as_tensor(_points(pred_pcd), device=dev)
    tgt = torch.as_tensor(_points(gt_pcd), device=dev)
    soff = torch.tensor([0, src.shape[0]], dtype=torch.int32, device=dev)
    toff = torch.tensor([0, tgt.shape[0]], dtype=torch.int32, device=dev)
    init = torch.eye(4, dtype=torch.float64, device=dev).reshape(1, 4, 4)
    T, moved, _ = ops.icp_p2p(src, soff, tgt, toff, init, th=threshold, max_iteration=max_iteration)
    return T[0].cpu().numpy(), PointCloud(moved.cpu().numpy())
