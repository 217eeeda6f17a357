"""Build-container-only helper: make /root/reference/PointCloud importable without the missing wheels.

The reference's hot-path modules import pytorch3d and open3d at module top (mlp_reg.py:4,13-14,
cluster_icp.py:1, dq_func.py:2, model_utils.py:5).  Neither wheel exists in this image, so stub
modules are placed in ``sys.modules`` whose bodies are the oracle's restatements.  What importing
the reference through these stubs pins is the REFERENCE'S OWN composition logic (operand order,
residuals, clamps, best-tracking, label bookkeeping) -- not the third-party arithmetic.

Used only by make_golden.py.  Never imported by tests that run on the GPU box (/root/reference
does not exist there).
"""
import sys
import types

import numpy as np
import torch

REF = "/root/reference/PointCloud"


class _Vec(np.ndarray):
    pass


class _PointCloud:
    def __init__(self, pts=None):
        self.points = np.zeros((0, 3)) if pts is None else np.asarray(pts, np.float64)

    def paint_uniform_color(self, c):
        return self

    def transform(self, T):
        T = np.asarray(T)
        self.points = self.points @ T[:3, :3].T + T[:3, 3]
        return self


def install():
    sys.path.insert(0, "/root/repo")
    from oracle import chamfer, icp, transforms

    p3d = types.ModuleType("pytorch3d")
    tr = types.ModuleType("pytorch3d.transforms")
    for name in ("quaternion_to_matrix", "matrix_to_quaternion", "quaternion_raw_multiply",
                 "quaternion_invert", "matrix_to_euler_angles", "euler_angles_to_matrix",
                 "matrix_to_rotation_6d", "rotation_6d_to_matrix"):
        setattr(tr, name, getattr(transforms, name))
    loss = types.ModuleType("pytorch3d.loss")
    loss.chamfer_distance = chamfer.chamfer_distance
    p3d.transforms, p3d.loss = tr, loss
    sys.modules.update({"pytorch3d": p3d, "pytorch3d.transforms": tr, "pytorch3d.loss": loss})

    o3d = types.ModuleType("open3d")
    class _Mesh:                                     # coordinate-frame gizmos the reference builds for its viewer calls
        def transform(self, T):
            return self

        def paint_uniform_color(self, c):
            return self

    o3d.geometry = types.SimpleNamespace(PointCloud=_PointCloud, TriangleMesh=types.SimpleNamespace(
        create_coordinate_frame=lambda size=1.0, origin=None: _Mesh()))
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.asarray(a, np.float64))

    class _Result:
        pass

    def registration_icp(source, target, th, init, estimation=None, criteria=None):
        r = _Result()
        r.transformation, r.fitness, r.inlier_rmse, _ = icp.registration_icp(
            source.points, target.points, th, init, max_iteration=criteria.max_iteration)
        return r

    reg = types.SimpleNamespace(
        registration_icp=registration_icp,
        TransformationEstimationPointToPoint=lambda: None,
        ICPConvergenceCriteria=lambda max_iteration=30: types.SimpleNamespace(max_iteration=max_iteration))
    o3d.pipelines = types.SimpleNamespace(registration=reg)
    def read_point_cloud(path):
        """ascii PLY with x y z vertex properties only (what tests/_ply.write_ascii_ply writes)."""
        with open(path) as f:
            lines = f.read().split("\n")
        end = lines.index("end_header")
        n = int([l for l in lines[:end] if l.startswith("element vertex")][0].split()[2])
        return _PointCloud(np.array([[float(v) for v in l.split()[:3]] for l in lines[end + 1:end + 1 + n]], np.float64))

    o3d.io = types.SimpleNamespace(read_point_cloud=read_point_cloud)
    o3d.visualization = types.SimpleNamespace(draw_geometries=lambda *a, **k: None)
    sys.modules["open3d"] = o3d

    # torch >= 2.7 dropped ReduceLROnPlateau(verbose=...) which mlp_reg.py:49 still passes
    base = torch.optim.lr_scheduler.ReduceLROnPlateau

    class _Plateau(base):
        def __init__(self, *a, verbose=None, **k):
            super().__init__(*a, **k)

    torch.optim.lr_scheduler.ReduceLROnPlateau = _Plateau
    if REF not in sys.path:
        sys.path.insert(0, REF)
