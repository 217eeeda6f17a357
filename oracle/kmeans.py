"""Lloyd k-means with sklearn.cluster.k_means(X, init=<ndarray>, n_init=1) semantics (oracle).

Reference call sites: PointCloud/mlp_reg.py:204 (resample_cluster), cluster_icp.py:67.  Recipe:
sklearn/cluster/_kmeans.py:1479-1484 (centre by mean), :279-288 (tol), :699-750 (loop, strict
convergence, final E-step), _k_means_lloyd.pyx:191-218 (first-min argmin, accumulation),
_k_means_common.pyx:167-311 (empty-cluster relocation, averaging by multiplication with 1/w,
centre shift).  Arithmetic lives in creg_oracle.c (explicit fma order shared with the HIP kernel).
"""
import numpy as np

from ._clib import lib


def assign(X: np.ndarray, C: np.ndarray) -> np.ndarray:
    X = np.ascontiguousarray(X, np.float64)
    C = np.ascontiguousarray(C, np.float64)
    lab = np.empty(len(X), np.int32)
    lib().oracle_kmeans_assign_f64(X.ctypes.data, len(X), C.ctypes.data, len(C), lab.ctypes.data)
    return lab


def k_means(X: np.ndarray, init: np.ndarray, n_clusters: int = None, n_init: int = 1,
            max_iter: int = 300, tol: float = 1e-4):
    """Returns (centers (K,3) f64, labels (N,) int32, inertia f64, n_iter) -- sklearn's tuple + n_iter."""
    X = np.ascontiguousarray(X, np.float64)
    C0 = np.ascontiguousarray(init, np.float64)
    k = len(C0)
    if n_clusters is not None and n_clusters != k:
        raise ValueError("init has %d rows, n_clusters=%d" % (k, n_clusters))
    centers = np.empty((k, 3), np.float64)
    labels = np.empty(len(X), np.int32)
    inertia = np.zeros(1, np.float64)
    n_iter = lib().oracle_kmeans_lloyd_f64(X.ctypes.data, len(X), C0.ctypes.data, k, max_iter, tol,
                                           centers.ctypes.data, labels.ctypes.data,
                                           inertia.ctypes.data)
    return centers, labels, float(inertia[0]), n_iter


def farthest_point_sample(X: np.ndarray, m: int) -> np.ndarray:
    X = np.ascontiguousarray(X, np.float64)
    sel = np.empty(m, np.int64)
    lib().oracle_fps_f64(X.ctypes.data, len(X), m, sel.ctypes.data)
    return sel
