"""Oracle for link refinement.  TEST INFRASTRUCTURE ONLY.

Restates reference PointCloud/link.py:85-127 ``refine_links_clusters`` (per time step and link:
registration_icp(link_t, link_first, threshold 1, identity, point-to-point, max_iteration 100000),
the source moved by the result) -- PINNED: the reference function runs under shims to produce
tests/golden/link_refine_reference.npz -- and the ICP filter + Chamfer of Sim/evaluation.py:69-81,358-362.
The ICP itself is ``oracle.icp.registration_icp`` (open3d restated; parity UNPINNED, see oracle/__init__.py).
"""
import numpy as np

from .icp import registration_icp


def refine_links(clusters_by_t, first, dof, th=1.0, max_iteration=100000):
    """clusters_by_t: list over time of lists of (M,3) link clouds; first: the link clouds at start_steps.
    Returns the moved clouds, same nesting (link.py:93-125)."""
    out = []
    for clusters in clusters_by_t:
        moved = []
        for _, c, f in zip(range(dof + 1), clusters, first):
            c = np.asarray(c, np.float64)
            T, _, _, _ = registration_icp(c, np.asarray(f, np.float64), th, np.eye(4), max_iteration)
            moved.append(c @ T[:3, :3].T + T[:3, 3])
        out.append(moved)
    return out


def icp_filter(pred, gt, th=0.01, max_iteration=20000):
    """Sim/evaluation.py:358-362: returns (T, pred moved by T)."""
    pred = np.asarray(pred, np.float64)
    T, _, _, _ = registration_icp(pred, np.asarray(gt, np.float64), th, np.eye(4), max_iteration)
    return T, pred @ T[:3, :3].T + T[:3, 3]
