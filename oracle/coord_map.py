"""Oracle for the pose-sequence distance maps.  TEST INFRASTRUCTURE ONLY.

Restates, in numpy fp64:

* reference ``CoordMap.coord_dist_map`` PointCloud/coord_map.py:230-307 (both ``diff`` branches),
  ``coord_dist_map_legacy`` :309-332, ``load_matrix`` :185-221 (pose -> xyz + quaternion) and
  ``get_scale`` :175-183                                   -- PINNED: the reference module is
  imported under shims by tests/golden/make_golden.py and its own loops produce the goldens;
* the four functions of **roma** (PyPI ``roma``, unpinned in the reference's requirements; call
  sites coord_map.py:14,262,267,290) those loops call: ``rotmat_to_rotvec`` (= ``rotmat_to_unitquat``
  adapted from SciPy's ``Rotation.from_matrix`` + ``unitquat_to_rotvec`` with the shortest arc),
  ``utils.rotvec_geodesic_distance`` (= ``unitquat_geodesic_distance`` of ``rotvec_to_unitquat``:
  ``4 asin(0.5 min(|q2 - q1|, |q2 + q1|))``) and ``rotmat_geodesic_distance`` (``acos`` of the clamped
  ``(trace(R1^T R2) - 1) / 2``).  The wheel is absent from this image and from /root/reference
  -- parity UNPINNED for this third-party arithmetic; cross-checked against
  ``scipy.spatial.transform.Rotation`` (tests/test_oracle_golden.py).
"""
import math

import numpy as np


# ------------------------------------------------------------------------------------------ roma
def rotmat_to_unitquat(R: np.ndarray) -> np.ndarray:
    """(...,3,3) -> (...,4) xyzw.  roma.mappings.rotmat_to_unitquat (SciPy's decision-matrix form)."""
    R = np.asarray(R, np.float64)
    m = R.reshape(-1, 3, 3)
    n = m.shape[0]
    dec = np.empty((n, 4))
    dec[:, :3] = np.diagonal(m, axis1=1, axis2=2)
    dec[:, 3] = dec[:, :3].sum(axis=1)
    choice = dec.argmax(axis=1)
    q = np.empty((n, 4))
    for r in range(n):
        c = choice[r]
        if c != 3:
            i, j, k = c, (c + 1) % 3, (c + 2) % 3
            q[r, i] = 1 - dec[r, 3] + 2 * m[r, i, i]
            q[r, j] = m[r, j, i] + m[r, i, j]
            q[r, k] = m[r, k, i] + m[r, i, k]
            q[r, 3] = m[r, k, j] - m[r, j, k]
        else:
            q[r, 0] = m[r, 2, 1] - m[r, 1, 2]
            q[r, 1] = m[r, 0, 2] - m[r, 2, 0]
            q[r, 2] = m[r, 1, 0] - m[r, 0, 1]
            q[r, 3] = 1 + dec[r, 3]
    q = q / np.linalg.norm(q, axis=1)[:, None]
    return q.reshape(R.shape[:-2] + (4,))


def unitquat_to_rotvec(q: np.ndarray) -> np.ndarray:
    """roma.mappings.unitquat_to_rotvec(shortest_arc=True)."""
    q = np.array(q, np.float64).reshape(-1, 4)
    q[q[:, 3] < 0] *= -1
    half = np.arctan2(np.linalg.norm(q[:, :3], axis=1), q[:, 3])
    angle = 2 * half
    small = np.abs(angle) <= 1e-3
    scale = np.empty(len(q))
    scale[small] = 2 + angle[small] ** 2 / 12 + 7 * angle[small] ** 4 / 2880
    scale[~small] = angle[~small] / np.sin(half[~small])
    return scale[:, None] * q[:, :3]


def rotmat_to_rotvec(R: np.ndarray) -> np.ndarray:
    R = np.asarray(R, np.float64)
    return unitquat_to_rotvec(rotmat_to_unitquat(R)).reshape(R.shape[:-2] + (3,))


def rotvec_to_unitquat(v: np.ndarray) -> np.ndarray:
    v = np.asarray(v, np.float64)
    flat = v.reshape(-1, 3)
    norms = np.linalg.norm(flat, axis=-1)
    small = norms <= 1e-3
    scale = np.empty(len(flat))
    scale[small] = 0.5 - norms[small] ** 2 / 48 + norms[small] ** 4 / 3840
    scale[~small] = np.sin(norms[~small] / 2) / norms[~small]
    q = np.empty((len(flat), 4))
    q[:, :3] = scale[:, None] * flat
    q[:, 3] = np.cos(norms / 2)
    return q.reshape(v.shape[:-1] + (4,))


def unitquat_geodesic_distance(q1, q2):
    q1, q2 = np.asarray(q1, np.float64), np.asarray(q2, np.float64)
    return 4 * np.arcsin(0.5 * np.minimum(np.linalg.norm(q2 - q1, axis=-1), np.linalg.norm(q2 + q1, axis=-1)))


def rotvec_geodesic_distance(v1, v2):
    return unitquat_geodesic_distance(rotvec_to_unitquat(v1), rotvec_to_unitquat(v2))


def rotmat_geodesic_distance(R1, R2, clamping=1.0):
    R = np.swapaxes(np.asarray(R1, np.float64), -1, -2) @ np.asarray(R2, np.float64)
    cos = 0.5 * (R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2] - 1.0)
    return np.arccos(np.clip(cos, -clamping, clamping))


# ------------------------------------------------------------------------------------------ CoordMap
def coords_from_matrices(matrices: np.ndarray) -> np.ndarray:
    """load_matrix coord_map.py:204-219: (T,K,4,4) -> (T,K,7) = xyz + pytorch3d real-first quaternion."""
    import torch
    from . import transforms
    M = np.asarray(matrices)
    q = transforms.matrix_to_quaternion(torch.as_tensor(M[..., :3, :3])).numpy()
    return np.concatenate([M[..., :3, 3], q], axis=-1)


def get_scale(coords: np.ndarray) -> float:
    return float(max(np.max(coords[0, :, i]) - np.min(coords[0, :, i]) for i in range(3)))


def coord_dist_map(matrices: np.ndarray, bounding_box: float, diff: bool = True):
    """coord_map.py:230-307.  matrices (T,K,4,4) -> (coord_dist_map (K,K,T') , sum_map (K,K)), T' = T-1 (diff) or T."""
    M = np.asarray(matrices, np.float64)
    T, K = M.shape[:2]
    lam_rot, lam_bbox = 1 / math.pi, 1 / (bounding_box * 2)
    xyz, R = M[:, :, :3, 3], M[:, :, :3, :3]
    maps = []
    if diff:
        trans_diff = np.diff(xyz, axis=0)                                        # (T-1,K,3)
        rel = np.swapaxes(R[:-1], -1, -2) @ R[1:]                                # R_i^T R_{i+1}
        rot_diff = rotmat_to_rotvec(rel)                                         # (T-1,K,3)
        for i in range(T - 1):
            d_xyz = lam_bbox * np.linalg.norm(trans_diff[i][:, None] - trans_diff[i][None], axis=-1)
            d_rpy = lam_rot * rotvec_geodesic_distance(rot_diff[i][:, None], rot_diff[i][None])
            trans_dist = np.linalg.norm(d_xyz[:, None] - d_xyz[None], axis=-1)   # distance between ROWS
            rot_dist = np.linalg.norm(d_rpy[:, None] - d_rpy[None], axis=-1)
            maps.append(trans_dist + rot_dist)
    else:
        for i in range(T):
            d_xyz = lam_bbox * np.linalg.norm(xyz[i][:, None] - xyz[i][None], axis=-1)
            d_rpy = lam_rot * rotmat_geodesic_distance(R[i][:, None], R[i][None])
            maps.append(d_xyz + d_rpy)
    cmap = np.stack(maps, axis=2)
    return cmap, np.sum(np.abs(cmap), axis=2)


def coord_dist_map_legacy(coords: np.ndarray):
    """coord_map.py:309-332: xyz relative to step 0 and the remaining pose coordinates, plain Euclidean
    distance matrices per step, summed map min-max normalised."""
    from scipy.spatial import distance_matrix
    maps = []
    for i in range(coords.shape[0]):
        xyz_i = coords[i][:, :3] - coords[0][:, :3]
        maps.append(distance_matrix(xyz_i, xyz_i) + distance_matrix(coords[i][:, 3:], coords[i][:, 3:]))
    cmap = np.stack(maps, axis=2)
    s = np.sum(np.abs(cmap), axis=2)
    return cmap, (s - np.min(s)) / (np.max(s) - np.min(s))
