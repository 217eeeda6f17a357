"""Farthest-point down-sampling on the GPU (open3d PointCloud.farthest_point_down_sample, reference
cluster_icp.py:43)."""
import ctypes

import numpy as np
import torch

from . import _lib, ops


def farthest_point_sample(points, num_samples: int) -> np.ndarray:
    """points (N,3) array-like or CUDA tensor (fp64) -> int64 indices of the selected points, in selection order."""
    L = _lib.load()
    if isinstance(points, torch.Tensor) and points.is_cuda:
        X = points.to(torch.float64).contiguous()
    else:
        X = torch.as_tensor(np.asarray(points, np.float64), device=_lib.device()).contiguous()
    n = X.shape[0]
    if not (1 <= num_samples <= n):
        raise ValueError("need 1 <= num_samples <= number of points")
    sel = torch.empty(num_samples, dtype=torch.int64, device=X.device)
    scratch = torch.empty(L.creg_fps_scratch_bytes(n), dtype=torch.uint8, device=X.device)
    _lib.check(L.creg_fps_f64(ops._p(X), n, num_samples, ops._p(sel), ops._p(scratch), ops._stream()), "creg_fps_f64")
    return sel.cpu().numpy()
