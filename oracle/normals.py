"""Oracle for the `--normal` branch (test infrastructure): numpy / scipy restatement of open3d 0.18's

    PointCloud.estimate_normals(KDTreeSearchParamHybrid(radius, max_nn))    (reference mlp_reg.py:191, cluster_icp.py:51)
    PointCloud.orient_normals_consistent_tangent_plane(k)                   (mlp_reg.py:192, cluster_icp.py:52)

open3d is not vendored by the reference and is absent from this image: PARITY UNPINNED -- what is restated is the published
algorithm (hybrid search = the max_nn nearest with squared distance < radius^2, the query point included; covariance from the
raw moments; normal = eigenvector of the smallest eigenvalue, (0,0,1) below three neighbours; orientation = Hoppe's consistent
tangent planes over the Euclidean MST of the Delaunay edges + kNN edges, weights 1 - |n_i . n_j|, Kruskal, traversal from the
highest point turned towards +z).  Written independently of autourdf_amd/normals.py: brute-force distances + argsort,
numpy.linalg.eigh, an explicit Kruskal with union-find and an explicit queue.
"""
import numpy as np


def hybrid_neighbours(P, radius, max_nn):
    d2 = ((P[:, None, :] - P[None, :, :]) ** 2).sum(-1)
    order = np.lexsort((np.broadcast_to(np.arange(len(P)), d2.shape), d2), axis=1)[:, :max_nn]      # by (distance, index)
    d_sorted = np.take_along_axis(d2, order, 1)
    ok = d_sorted < radius * radius if radius > 0 else np.ones_like(d_sorted, bool)
    return [order[i][ok[i]] for i in range(len(P))]


def estimate_normals(P, radius=0.1, max_nn=30):
    P = np.asarray(P, np.float64)
    out = np.tile([0.0, 0.0, 1.0], (len(P), 1))
    for i, nb in enumerate(hybrid_neighbours(P, radius, max_nn)):
        if len(nb) < 3:
            continue
        q = P[nb]
        m1 = q.sum(0) / len(nb)
        m2 = (q[:, :, None] * q[:, None, :]).sum(0) / len(nb)
        cov = m2 - np.outer(m1, m1)
        w, v = np.linalg.eigh(cov)
        out[i] = v[:, 0] / np.linalg.norm(v[:, 0])
    return out


def _kruskal(n, edges, w):
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    tree = []
    for j in np.argsort(w, kind="stable"):
        a, b = find(int(edges[j, 0])), find(int(edges[j, 1]))
        if a != b:
            parent[a] = b
            tree.append((int(edges[j, 0]), int(edges[j, 1])))
    return tree


def orient_normals_consistent_tangent_plane(P, normals, k=30):
    from scipy.spatial import Delaunay
    P = np.asarray(P, np.float64)
    N = np.array(normals, np.float64)
    n = len(P)
    tet = Delaunay(P).simplices
    e = np.unique(np.sort(np.concatenate([tet[:, [a, b]] for a in range(4) for b in range(a + 1, 4)]), axis=1), axis=0)
    emst = _kruskal(n, e, ((P[e[:, 0]] - P[e[:, 1]]) ** 2).sum(1))
    knn = hybrid_neighbours(P, -1.0, k)
    extra = [(i, int(j)) for i in range(n) for j in knn[i] if j != i]
    edges = np.unique(np.sort(np.array(emst + extra), axis=1), axis=0)
    w = 1.0 - np.abs((N[edges[:, 0]] * N[edges[:, 1]]).sum(1))
    adj = [[] for _ in range(n)]
    for a, b in _kruskal(n, edges, w):
        adj[a].append(b); adj[b].append(a)
    start = int(np.argmax(P[:, 2]))
    if N[start, 2] < 0:
        N[start] = -N[start]
    seen = np.zeros(n, bool)
    seen[start] = True
    queue = [start]
    while queue:
        v = queue.pop(0)
        for u in adj[v]:
            if not seen[u]:
                seen[u] = True
                if N[u] @ N[v] < 0:
                    N[u] = -N[u]
                queue.append(u)
    return N


def point_features(P, radius=0.1, max_nn=30, k=30, scale=0.5):
    P = np.asarray(P, np.float64)
    N = orient_normals_consistent_tangent_plane(P, estimate_normals(P, radius, max_nn), k)
    return np.hstack([P, scale * N]), N
