"""Drop-in for the data side of the reference's ``PointCloud/coord_map.py``: the ``CoordMap`` that
consumes ``match()``'s ``matrix/*.npy`` / ``cluster/*.npz`` output.

Same constructor, attributes and method signatures as the reference class (coord_map.py:130-332):
``coords`` (T,K,7), ``matrices`` (T,K,4,4), ``clusters``, ``num_coords``, ``scale``, ``bounding_box``;
``load_matrix``, ``load_cluster``, ``get_scale``, ``get_bounding_box``, ``coord_dist_map(diff=True)``,
``coord_dist_map_legacy``.  The O(T K^2)/O(T K^3) Python loops with per-element torch / roma calls
(:250-301) are one launch of ``creg_coord_dist_map_f64``; the pose -> quaternion loop of
``load_matrix`` (:204-219) one launch of ``creg_pose_coords_f64``.  Everything is evaluated in fp64
(the reference's arrays are float64 whenever frame 0 -- saved as float64, mlp_reg.py:257-263 -- is in
the range; an all-float32 range is promoted, which only removes the reference's own float32 rounding).

The graph / URDF half of the reference file (MST, silhouette clustering, joints, GUI) is out of scope.
There is no CPU fallback: the methods need the HIP library and a GPU.
"""
import glob

import numpy as np
import torch

from . import _lib, ops
from .cluster_icp import read_point_cloud


class CoordMap:
    """Coordinate correlation map (reference coord_map.py:130-150)."""

    def __init__(self, data_path, raw_path, gt_data=False, start_steps=0, end_steps=0):
        self.data_path = data_path
        self.gt_data = gt_data
        self.start_steps = start_steps
        self.end_steps = end_steps
        self.coords, self.matrices = self.load_matrix(start_steps, end_steps)
        self.clusters = self.load_cluster(start_steps, end_steps)
        self.num_coords = self.coords.shape[1]
        self.scale = self.get_scale()
        self.bounding_box = self.get_bounding_box(raw_path)

    @classmethod
    def from_arrays(cls, matrices, bounding_box, clusters=None):
        """Poses already in memory ((T,K,4,4) array or device tensor): no file round trip."""
        self = cls.__new__(cls)
        self.data_path, self.gt_data, self.start_steps, self.end_steps = None, False, 0, 0
        self._M = torch.as_tensor(matrices, dtype=torch.float64).to(_lib.device(matrices)).contiguous()
        self.matrices = self._M.cpu().numpy()
        self.coords = ops.pose_coords(self._M).cpu().numpy()
        self.clusters = clusters if clusters is not None else []
        self.num_coords = self.coords.shape[1]
        self.scale = self.get_scale()
        self.bounding_box = float(bounding_box)
        return self

    # ---- loading (coord_map.py:185-228) -------------------------------------------------------------
    def load_matrix(self, start_steps=0, end_steps=0):
        files = sorted(glob.glob(self.data_path + 'matrix/*.npy'))[start_steps:end_steps]
        if not files:
            raise FileNotFoundError(f"no matrix/*.npy under {self.data_path} in [{start_steps}:{end_steps}]")
        matrices = np.array([np.load(f) for f in files])           # (T,K,4,4); float64 as soon as one file is
        self._M = torch.as_tensor(matrices, dtype=torch.float64).to(_lib.device()).contiguous()
        coords = ops.pose_coords(self._M).cpu().numpy().astype(matrices.dtype, copy=False)
        return coords, matrices

    def load_cluster(self, start_steps=0, end_steps=0):
        files = sorted(glob.glob(self.data_path + 'cluster/*.npz'))[start_steps:end_steps]
        return [np.load(f) for f in files]

    def get_scale(self):
        return float(max(np.max(self.coords[0, :, i]) - np.min(self.coords[0, :, i]) for i in range(3)))

    def get_bounding_box(self, raw_path):
        """Diagonal of the AABB of every raw frame of the sequence (coord_map.py:153-173)."""
        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        for path in sorted(glob.glob(str(raw_path) + '*/')):
            pts = np.asarray(read_point_cloud(path + 'robot.ply').points)
            if len(pts):
                lo, hi = np.minimum(lo, pts.min(0)), np.maximum(hi, pts.max(0))
        if not np.all(np.isfinite(lo)):
            raise FileNotFoundError(f"no */robot.ply under {raw_path}")
        return float(np.linalg.norm(hi - lo))

    # ---- the maps (coord_map.py:230-332) ------------------------------------------------------------
    def coord_dist_map(self, diff=True):
        """num_seg x num_seg x time-step matrix and its sum over time, numpy arrays like the reference."""
        d_map, s_map = ops.coord_dist_map(self._M, self.bounding_box, diff)
        return d_map.cpu().numpy(), s_map.cpu().numpy()

    def coord_dist_map_legacy(self, diff=True):
        """xyz relative to step 0 + remaining pose coordinates, Euclidean distance matrices per step
        (coord_map.py:309-332); `diff` is ignored there too.  Two cdist calls per step on the device."""
        c = torch.as_tensor(np.asarray(self.coords, np.float64), device=_lib.device(getattr(self, "_M", None)))
        xyz = c[:, :, :3] - c[:1, :, :3]
        mode = "donot_use_mm_for_euclid_dist"          # exact differences, not the |a|^2 + |b|^2 - 2ab expansion
        rest = c[:, :, 3:].contiguous()
        d = torch.cdist(xyz, xyz, compute_mode=mode) + torch.cdist(rest, rest, compute_mode=mode)
        cmap = d.permute(1, 2, 0).contiguous()
        s = cmap.abs().sum(dim=2)
        s = (s - s.min()) / (s.max() - s.min())
        return cmap.cpu().numpy(), s.cpu().numpy()
