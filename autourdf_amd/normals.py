"""Point normals for the reference's ``--normal`` branch (PointCloud/mlp_reg.py:190-192, cluster_icp.py:49-51):

    pc.estimate_normals(search_param=o3d.geometry.KDTreeSearchParamHybrid(radius=0.1, max_nn=30))
    pc.orient_normals_consistent_tangent_plane(30)

open3d (0.18 in the reference's environment) is not vendored and is absent here, so this is its PUBLISHED behaviour, restated:
``estimate_normals`` runs on the GPU (csrc/normals.hip: exhaustive hybrid neighbour search, raw-moment covariance, smallest
eigenvector); the orientation pass -- Hoppe et al.'s consistent tangent planes as open3d implements them: a Riemannian graph
(Euclidean minimum spanning tree of the Delaunay edges + k-nearest-neighbour edges) weighted 1 - |n_i . n_j|, its minimum
spanning tree, a traversal from the highest point (whose normal is turned towards +z) flipping every normal that disagrees
with its tree parent -- is a dependent graph walk over a Delaunay tetrahedralisation and stays on the host (scipy's Qhull
binding and sparse-graph routines; open3d itself runs it single-threaded on the CPU, with Qhull too).  The k nearest
neighbours it needs come from the same GPU search.  Parity with open3d is UNPINNED.
"""
import numpy as np
import torch

from . import _lib, ops


def estimate_normals(points, radius=0.1, max_nn=30):
    """open3d PointCloud.estimate_normals with KDTreeSearchParamHybrid(radius, max_nn): (n,3) float64 unit normals, signs
    as the eigen-solver leaves them."""
    X = torch.as_tensor(np.asarray(points), dtype=torch.float64, device=_lib.device()).contiguous()
    n, _, _ = ops.knn_normals(X, radius, max_nn)
    return n.cpu().numpy()


def _tree_order(n, rows, cols, w, start):
    """Minimum spanning forest of the weighted graph, then (order, parent) of a breadth-first walk from `start` over ITS component
    (open3d walks one component too: where the Riemannian graph of a cloud is disconnected, the other components keep the signs the
    eigen-solver gave them -- they do not appear in `order`)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import breadth_first_order, minimum_spanning_tree
    g = coo_matrix((w + 1.0, (rows, cols)), shape=(n, n)).tocsr()       # + 1: an MST is invariant under it, and a zero weight
    mst = minimum_spanning_tree(g)                                       #      (parallel normals) would read as "no edge"
    mst = mst + mst.T
    order, pred = breadth_first_order(mst, start, directed=False, return_predecessors=True)
    return order, pred


def orient_normals_consistent_tangent_plane(points, normals, k=30):
    """open3d PointCloud.orient_normals_consistent_tangent_plane(k): returns the re-oriented copy of `normals`."""
    from scipy.spatial import Delaunay
    P = np.asarray(points, np.float64)
    N = np.array(normals, np.float64)
    n = len(P)
    if n < 5:
        return N
    # Euclidean minimum spanning tree over the Delaunay edges
    tet = Delaunay(P).simplices
    e = np.concatenate([tet[:, [a, b]] for a in range(4) for b in range(a + 1, 4)])
    e = np.unique(np.sort(e, axis=1), axis=0)
    d2 = ((P[e[:, 0]] - P[e[:, 1]]) ** 2).sum(1)
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import minimum_spanning_tree
    emst = minimum_spanning_tree(coo_matrix((d2 + 1.0, (e[:, 0], e[:, 1])), shape=(n, n)).tocsr()).tocoo()
    edges = np.stack([emst.row, emst.col], 1)
    # + the k nearest neighbours of every point (GPU search; the first entry is the point itself)
    X = torch.as_tensor(P, device=_lib.device()).contiguous()
    _, idx, _ = ops.knn_normals(X, -1.0, min(k, 32), want_normals=False, want_idx=True)
    idx = idx.cpu().numpy()
    src = np.repeat(np.arange(n), idx.shape[1])
    dst = idx.reshape(-1)
    keep = (dst >= 0) & (dst != src)
    edges = np.concatenate([edges, np.stack([src[keep], dst[keep]], 1)])
    edges = np.unique(np.sort(edges, axis=1), axis=0)
    w = 1.0 - np.abs((N[edges[:, 0]] * N[edges[:, 1]]).sum(1))
    start = int(np.argmax(P[:, 2]))
    order, pred = _tree_order(n, edges[:, 0], edges[:, 1], w, start)
    if N[start, 2] < 0:
        N[start] = -N[start]
    for v in order[1:]:                                   # parents come before their children in a breadth-first order
        if (N[pred[v]] * N[v]).sum() < 0:
            N[v] = -N[v]
    return N


def point_features(points, radius=0.1, max_nn=30, k=30, scale=0.5):
    """[xyz | scale * oriented normal] (n,6) float64: what the reference feeds k_means under --normal (mlp_reg.py:199-202)."""
    P = np.asarray(points, np.float64)
    N = orient_normals_consistent_tangent_plane(P, estimate_normals(P, radius, max_nn), k)
    return np.hstack([P, N * scale]), N
