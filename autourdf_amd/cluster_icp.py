"""Frame loading, frame-0 segmentation and masked per-cluster ICP on the MI355X -- drop-in for
reference PointCloud/cluster_icp.py (``xyzrpy_to_matrix_scipy`` :7-12, ``Segments`` :14-115,
``masked_icp`` :118-191): same names, positional order, defaults, attributes and return values.

Open3D is not a dependency: ``robot.ply`` files are parsed here (ascii / binary little- or
big-endian, float or double vertices) and clouds are handed around as ``PointCloud`` objects
exposing ``.points`` like ``open3d.geometry.PointCloud`` does.  GUI calls of the reference
(``draw_geometries``, cluster_icp.py:106,183) are not reproduced.
"""
from glob import glob

import numpy as np
import torch
from scipy.spatial.transform import Rotation as R

from . import _lib, ops


def kmeans_plusplus_sklearn(X, k, random_state):
    """k-means++ seeding with scikit-learn's draw sequence (sklearn/cluster/_kmeans.py:174-275 as reached by
    ``k_means(X, init="k-means++", n_clusters=k)``, reference cluster_icp.py:67): on the mean-centred data, first
    centre by ``random_state.choice``, then 2 + log(k) candidates per centre from ``random_state.uniform`` against the
    cumulative closest-distance potential, the candidate with the smallest new potential wins.  ``random_state`` is a
    ``numpy.random.RandomState``; the reference passes none, i.e. numpy's GLOBAL one -- so ``np.random.seed(s)`` before
    ``Segments.k_means_cluster`` pins the reference and this function to the same seeds.  Host RNG by nature (a
    dependent chain of k draws); returns the indices of the chosen points."""
    X = np.asarray(X, np.float64)
    n = len(X)
    Xc = X - X.mean(axis=0)
    xx = np.einsum("ij,ij->i", Xc, Xc)

    def dist2(rows):                                  # sklearn's euclidean_distances(squared=True): |y|^2 - 2 x.y + |x|^2
        d = -2.0 * (Xc[rows] @ Xc.T)
        d += xx[rows][:, None]
        d += xx[None, :]
        np.maximum(d, 0, out=d)                       # (no zeroing of a candidate's own entry: sklearn 1.5-1.7 fills the diagonal only when X is Y,
        return d                                      #  which X[cand] is not -- the rounding residue stays in the potentials, as there)

    trials = 2 + int(np.log(k))
    w = np.ones(n)
    idx = np.full(k, -1, dtype=int)
    idx[0] = random_state.choice(n, p=w / w.sum())
    closest = dist2(idx[:1])[0]
    pot = closest @ w
    for c in range(1, k):
        rand_vals = random_state.uniform(size=trials) * pot
        cand = np.searchsorted(np.cumsum(w * closest, dtype=np.float64), rand_vals)
        np.clip(cand, None, n - 1, out=cand)
        d = dist2(cand)
        np.minimum(closest, d, out=d)
        pots = (d @ w.reshape(-1, 1)).ravel()
        best = int(np.argmin(pots))
        pot, closest, idx[c] = pots[best], d[best], cand[best]
    return idx


def xyzrpy_to_matrix_scipy(xyz, rpy):
    T = np.eye(4)
    T[:3, 3] = xyz
    T[:3, :3] = R.from_euler("xyz", rpy).as_matrix()
    return T


class PointCloud:
    """Minimal stand-in for open3d.geometry.PointCloud: ``.points`` is an (N,3) float64 array."""

    def __init__(self, points=None):
        self.points = np.zeros((0, 3)) if points is None else np.asarray(points, np.float64).reshape(-1, 3)
        self.colors = None

    def paint_uniform_color(self, c):
        self.colors = np.tile(np.asarray(c, np.float64), (len(self.points), 1))
        return self

    def transform(self, T):
        T = np.asarray(T, np.float64)
        self.points = self.points @ T[:3, :3].T + T[:3, 3]
        return self

    def farthest_point_down_sample(self, num_samples):
        from .fps import farthest_point_sample
        return PointCloud(self.points[farthest_point_sample(self.points, num_samples)])


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_point_cloud(path) -> PointCloud:
    """Vertex x/y/z of a PLY file as float64 (what o3d.io.read_point_cloud yields, cluster_icp.py:41)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise IOError(f"{path}: not a PLY file")
        fmt, n_vert, props, in_vertex = None, 0, [], False
        before = []                  # elements declared ahead of `vertex`: [count, [property dtypes] or None if it has a list]
        seen_vertex = False
        while True:
            line = f.readline()
            if not line:
                raise IOError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vert, seen_vertex = int(tok[2]), True
                elif not seen_vertex:
                    before.append([int(tok[2]), []])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise IOError(f"{path}: list property on vertices is not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "property" and not seen_vertex and before:
                if tok[1] == "list":
                    before[-1][1] = None
                elif before[-1][1] is not None:
                    before[-1][1].append(_PLY_TYPES[tok[1]])
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        if not all(a in names for a in "xyz"):
            raise IOError(f"{path}: vertex element lacks x/y/z")
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise IOError(f"{path}: unknown or missing PLY format {fmt!r}")
        # the vertex records start after every element declared before `vertex`: skip them (fixed-size records only)
        for count, types in before:
            if types is None:
                raise IOError(f"{path}: an element with list properties precedes the vertices; cannot locate them")
            if fmt == "ascii":
                for _ in range(count):
                    f.readline()
            else:
                f.seek(count * sum(np.dtype(t).itemsize for t in types), 1)
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=n_vert, ndmin=2)
            pts = np.stack([data[:, names.index(a)] for a in "xyz"], 1)
        else:
            order = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(n, order + t) for n, t in props])
            raw = f.read(dt.itemsize * n_vert)
            if len(raw) != dt.itemsize * n_vert:
                raise IOError(f"{path}: truncated vertex data")
            rec = np.frombuffer(raw, dtype=dt, count=n_vert)
            pts = np.stack([rec[a].astype(np.float64) for a in "xyz"], 1)
    return PointCloud(pts)


class Segments:
    """Frames of one sequence + the frame-0 segmentation (reference cluster_icp.py:14-115)."""

    def __init__(self, data_path, sample_size=None) -> None:
        self.pc_path = data_path
        self.pc_list = []
        self.init_coord_list = []
        self.init_matrix_list = []
        self.init_segment_list = []
        self.data_size = 0
        self.sample_size = sample_size
        self._load_pc()

    def _load_pc(self):
        sub_path = sorted(glob(self.pc_path + "*/"))
        self.data_size = len(sub_path)
        print(f"Found {self.data_size} sub path")
        for path in sub_path:
            pc = read_point_cloud(path + "robot.ply")
            if self.sample_size is not None:
                pc = pc.farthest_point_down_sample(self.sample_size)
            self.pc_list.append(pc)

    @classmethod
    def from_arrays(cls, frames):
        """Build from in-memory (N,3) arrays (synthetic sequences) instead of a directory of PLYs."""
        self = cls.__new__(cls)
        self.pc_path, self.sample_size = None, None
        self.pc_list = [PointCloud(f) for f in frames]
        self.init_coord_list, self.init_matrix_list, self.init_segment_list = [], [], []
        self.data_size = len(self.pc_list)
        return self

    def k_means_cluster(self, pc_id=0, num=30, normal=False, colors=None, seed=None):
        """k-means++ seeding (host RNG, scikit-learn's draw sequence) + Lloyd on the GPU (K2); then per
        cluster a frame with R = I at the centroid and the points in that frame
        (cluster_icp.py:86-99).  Like the reference, the draw comes from numpy's GLOBAL RandomState unless
        ``seed`` is given (``np.random.seed(s)`` pins both implementations to the same segmentation)."""
        _lib.load()
        dev = _lib.device()
        pc_np = np.asarray(self.pc_list[pc_id].points)
        rs = np.random.RandomState(seed) if seed is not None else np.random.mtrand._rand
        X = torch.as_tensor(pc_np, dtype=torch.float64, device=dev)
        if normal:
            # cluster_icp.py:49-62: normals (hybrid radius 0.1 / 30 neighbours, consistently oriented), k-means++ and Lloyd
            # over [xyz | 0.5 n]; everything after the labels uses xyz only
            from .normals import point_features
            feat, nrm = point_features(pc_np)
            self.pc_list[pc_id].normals = nrm
            init6 = feat[kmeans_plusplus_sklearn(feat, num, rs)]
            _, labels, _, _ = ops.kmeans_lloyd_nd(torch.as_tensor(feat, dtype=torch.float64, device=dev).contiguous(),
                                                  torch.as_tensor(init6, dtype=torch.float64, device=dev).contiguous())
        else:
            init = pc_np[kmeans_plusplus_sklearn(pc_np, num, rs)]        # data points: centring them in the kernel is exact
            _, labels, _, _ = ops.kmeans_lloyd(X, torch.as_tensor(init, dtype=torch.float64, device=dev))
        lab = labels.long()
        counts = torch.zeros(num, dtype=torch.float64, device=dev).index_add_(0, lab, torch.ones_like(X[:, 0]))
        # np.mean over the cluster's own points (cluster_icp.py:86), not the Lloyd centre
        grouped, off = ops.group_to_local(X, labels, torch.eye(4, dtype=torch.float64, device=dev).repeat(num, 1, 1))
        off_h = off.cpu().numpy()
        grouped_h = grouped.cpu().numpy()
        new_pcd = []
        for i in range(num):
            pts = grouped_h[off_h[i]:off_h[i + 1]]
            center = np.mean(pts, axis=0)
            xyzrpy = np.array([center[0], center[1], center[2], 0.0, 0.0, 0.0])
            matrix = xyzrpy_to_matrix_scipy(xyzrpy[:3], xyzrpy[3:])
            self.init_coord_list.append(xyzrpy)
            self.init_matrix_list.append(matrix)
            inv = np.linalg.inv(matrix)
            self.init_segment_list.append((inv @ np.hstack([pts, np.ones((len(pts), 1))]).T)[:3].T)
            pcd = PointCloud(pts)
            pcd.paint_uniform_color(colors[i] if colors is not None else np.random.rand(3))
            new_pcd.append(pcd)
        return new_pcd

    def visualize(self, pcs):
        raise NotImplementedError("visualisation needs Open3D's GUI (out of scope)")


def masked_icp(clusters_local, clusters_world, step_pc_np, matrices, visual=False, ori=False, scale=1.2, th=1,
               colors=None, max_iteration=10000):
    """Per-cluster AABB-masked point-to-point ICP on the GPU (K4): one launch for all clusters.
    Returns (list of world-frame clusters (M_k,3) f64, new matrices (K,4,4) f64) like the reference."""
    if visual:
        raise NotImplementedError("visual=True needs Open3D's GUI (out of scope)")
    dev = _lib.device(step_pc_np, matrices, *clusters_local[:1])
    k = len(clusters_local)
    local, off = ops.pack_clusters(clusters_local, dev, torch.float64)
    # the two lists may be segmented differently: match() --mlp_icp hands the frame-0 clusters as sources and the
    # trained clouds of the current segmentation as boxes (mlp_reg.py:248,325; only min/max of c_world is used)
    world, off_w = ops.pack_clusters(clusters_world, dev, torch.float32)
    if len(clusters_world) != k or len(matrices) != k:
        raise ValueError("clusters_local, clusters_world and matrices must hold one entry per cluster")
    frame = torch.as_tensor(np.asarray(step_pc_np), dtype=torch.float64, device=dev).contiguous()
    M = torch.as_tensor(np.asarray(matrices), dtype=torch.float64, device=dev).contiguous()
    M_out, w_out, _ = ops.masked_icp(local, world, off, frame, M, scale, th, max_iteration, ori, world_offsets=off_w)
    off_h = off.cpu().numpy()
    w_h = w_out.cpu().numpy()
    return [w_h[off_h[i]:off_h[i + 1]] for i in range(k)], M_out.cpu().numpy()
