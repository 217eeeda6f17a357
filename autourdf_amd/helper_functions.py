"""npz cluster IO and pose <-> xyz+quaternion helpers (drop-in for reference
PointCloud/helper_functions.py:10-45; same names, same on-disk layout: keys '0'..'K-1')."""
import numpy as np
import torch


def save_pc_npz(segment_list, path):
    np.savez(path, **{str(i): np.asarray(pc) for i, pc in enumerate(segment_list)})


def load_pc_npz(path):
    with np.load(path) as z:
        return [z[k] for k in z.keys()]


def matrix2xyzquant_torch(matrix):
    """4x4 -> (x, y, z, qw, qx, qy, qz) (real-first quaternion, like the reference's output)."""
    from . import _lib, ops
    M = torch.as_tensor(matrix, dtype=torch.float32, device=_lib.device(matrix))
    q = ops.matrix_to_quat(M[:3, :3].reshape(1, 3, 3).contiguous())[0]
    return torch.cat([M[:3, 3], q])


def xyzquant2matrix_torch(xyzquat):
    from . import _lib, ops
    v = torch.as_tensor(xyzquat, dtype=torch.float32, device=_lib.device(xyzquat))
    out = torch.eye(4, device=v.device)
    out[:3, :3] = ops.quat_to_matrix(v[3:].reshape(1, 4).contiguous())[0]
    out[:3, 3] = v[:3]
    return out
