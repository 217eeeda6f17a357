"""Per-cluster masked point-to-point ICP, oracle (numpy fp64).

Restates reference PointCloud/cluster_icp.py:118-191 (AABB x scale mask with strict inequalities,
per-cluster ICP, optional keep-translation) and open3d==0.18.0
``pipelines.registration.registration_icp`` with ``TransformationEstimationPointToPoint`` and
``ICPConvergenceCriteria(max_iteration=...)`` (relative_fitness = relative_rmse = 1e-6):
nearest neighbour within ``th`` (brute force here instead of a KD-tree: same answer), fitness =
inliers/|source|, inlier_rmse, Umeyama/Kabsch without scaling (``Eigen::umeyama``: R = U S V^T,
S = diag(1,1,sign det(U)det(V))), update composed on the left.  open3d is not vendored: parity
UNPINNED; cross-checked on known rigid motions in tests/test_oracle_golden.py (test_kabsch_known_motion_reflection_and_planar, test_masked_icp_matches_reference).
"""
import numpy as np


def kabsch(src: np.ndarray, dst: np.ndarray, w: np.ndarray = None) -> np.ndarray:
    """4x4 rigid T minimising sum w |T src - dst|^2 (w = 1 by default); identity when there are no pairs."""
    T = np.eye(4)
    if len(src) == 0 or (w is not None and not np.sum(w) > 0):
        return T
    w = np.ones(len(src)) if w is None else np.asarray(w, np.float64)
    ms, md = (w[:, None] * src).sum(0) / w.sum(), (w[:, None] * dst).sum(0) / w.sum()
    sigma = ((dst - md) * w[:, None]).T @ (src - ms) / w.sum()
    U, _, Vt = np.linalg.svd(sigma)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    T[:3, :3], T[:3, 3] = R, md - R @ ms
    return T


_DENSE_PAIRS = 4_000_000                     # above this many (source, target) pairs the C loop replaces the dense numpy form


def _nn_l2_c(src_w: np.ndarray, tgt: np.ndarray):
    from ._clib import lib
    a = np.ascontiguousarray(src_w, np.float64)
    b = np.ascontiguousarray(tgt, np.float64)
    d2 = np.empty(len(a), np.float64)
    j = np.empty(len(a), np.int64)
    lib().oracle_nn_l2_f64(a.ctypes.data, len(a), b.ctypes.data, len(b), d2.ctypes.data, j.ctypes.data)
    return d2, j


def _correspond(src_w: np.ndarray, tgt: np.ndarray, th: float):
    if len(tgt) == 0 or len(src_w) == 0:
        return np.zeros(0, int), np.zeros(0, int), 0.0, 0.0
    if len(src_w) * len(tgt) <= _DENSE_PAIRS:
        d2 = ((src_w[:, None, :] - tgt[None, :, :]) ** 2).sum(-1)
        j = d2.argmin(1)
        dmin = d2[np.arange(len(src_w)), j]
    else:                                    # the same search without the dense (n, m, 3) array (creg_oracle.c: oracle_nn_l2_f64)
        dmin, j = _nn_l2_c(src_w, tgt)
    ok = dmin < th * th                      # strict: open3d KDTreeFlann::SearchHybrid keeps d^2 < r^2 (lower_bound on the sorted distances)
    i = np.nonzero(ok)[0]
    n = len(i)
    fitness = n / len(src_w)
    rmse = float(np.sqrt(dmin[ok].sum() / n)) if n else 0.0
    return i, j[ok], fitness, rmse


def registration_icp(source, target, th, init, max_iteration=10000, rel_fitness=1e-6, rel_rmse=1e-6):
    T = np.array(init, np.float64)
    src_w = source @ T[:3, :3].T + T[:3, 3]
    i, j, fit, rmse = _correspond(src_w, target, th)
    n_iter = 0
    for n_iter in range(1, max_iteration + 1):
        upd = kabsch(src_w[i], target[j])
        T = upd @ T
        src_w = src_w @ upd[:3, :3].T + upd[:3, 3]
        pf, pr = fit, rmse
        i, j, fit, rmse = _correspond(src_w, target, th)
        if abs(pf - fit) < rel_fitness and abs(pr - rmse) < rel_rmse:
            break
    return T, fit, rmse, n_iter


def aabb_mask(c_world: np.ndarray, pts: np.ndarray, scale: float = 1.2) -> np.ndarray:
    lo, hi = c_world.min(0), c_world.max(0)
    centre, size = np.mean(np.stack([lo, hi], 1), axis=1), hi - lo
    lo2, hi2 = centre - 0.5 * scale * size, centre + 0.5 * scale * size
    return np.all((pts > lo2) & (pts < hi2), axis=1)


def masked_icp(clusters_local, clusters_world, step_pc_np, matrices, ori=False, scale=1.2, th=1,
               max_iteration=10000):
    world, mats = [], []
    for c_local, c_world, M in zip(clusters_local, clusters_world, matrices):
        tgt = step_pc_np[aabb_mask(np.asarray(c_world), step_pc_np, scale)]
        T, _, _, _ = registration_icp(np.asarray(c_local, np.float64), tgt, th, M, max_iteration)
        if ori:
            T = T.copy()
            T[:3, 3] = np.asarray(M)[:3, 3]
        world.append(np.asarray(c_local, np.float64) @ T[:3, :3].T + T[:3, 3])
        mats.append(T)
    return world, np.array(mats)
