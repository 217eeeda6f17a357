"""Mint tests/golden/coord_map_reference.npz (run in the BUILD CONTAINER only).

    python tests/golden/make_golden_coord_map.py

Source of truth: the reference's own ``CoordMap`` (PointCloud/coord_map.py) imported under ref_shims,
with three more stubs because its module top imports wheels this image lacks: ``roma`` (bodies = the
oracle's restatement of the four roma functions the loops call), and the reference's GUI / meshing
modules ``compute_joints``, ``visualize``, ``link`` (never touched by the methods run here).  What
this pins is the reference's composition logic (loop structure, lambdas, row-distance step, stacking,
file loading order); the roma arithmetic itself stays "parity unpinned" (see oracle/coord_map.py).
Fixture = inputs + expected outputs only.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
from oracle import coord_map as ocm  # noqa: E402
from scipy.spatial.transform import Rotation  # noqa: E402

roma = types.ModuleType("roma")
roma.rotmat_to_rotvec = lambda R: torch.from_numpy(ocm.rotmat_to_rotvec(R.numpy()))
roma.rotmat_geodesic_distance = lambda a, b: torch.from_numpy(np.asarray(ocm.rotmat_geodesic_distance(a.numpy(), b.numpy())))
roma.utils = types.SimpleNamespace(
    rotvec_geodesic_distance=lambda a, b: torch.from_numpy(np.asarray(ocm.rotvec_geodesic_distance(a.numpy(), b.numpy()))))
sys.modules["roma"] = roma
for name, attrs in (("compute_joints", ("estimate_joint_axes_from_tree", "create_urdf", "visualize_urdf")),
                    ("visualize", ("visualize_kinematic_tree",)),
                    ("link", ("save_links", "refine_links_clusters", "visualize_links", "link_mesh"))):
    m = types.ModuleType(name)
    for a in attrs:
        setattr(m, a, None)
    sys.modules[name] = m
import coord_map as ref_cm  # noqa: E402  (reference)


def pose_sequence(T, K, seed):
    """K cluster poses drifting over T steps; some clusters share their motion (same link), one is static."""
    rng = np.random.default_rng(seed)
    M = np.tile(np.eye(4), (T, K, 1, 1))
    M[0, :, :3, :3] = Rotation.random(K, random_state=seed).as_matrix()
    M[0, :, :3, 3] = rng.uniform(-0.3, 0.3, size=(K, 3))
    group = rng.integers(0, max(2, K // 2), size=K)
    for t in range(1, T):
        steps = {g: (Rotation.from_rotvec(rng.normal(scale=0.08, size=3)).as_matrix(), rng.normal(scale=0.01, size=3))
                 for g in set(group)}
        for k in range(K):
            dR, dt = steps[group[k]] if group[k] else (np.eye(3), np.zeros(3))
            M[t, k, :3, :3] = dR @ M[t - 1, k, :3, :3]
            M[t, k, :3, 3] = dR @ M[t - 1, k, :3, 3] + dt
    # what the files hold (mlp_reg.py:257-263,377): frame 0 float64, later frames float32
    M[1:] = M[1:].astype(np.float32).astype(np.float64)
    return M


def main():
    out = {}
    for tag, T, K, seed in (("a", 6, 7, 0), ("b", 4, 12, 1)):
        M = pose_sequence(T, K, seed)
        with tempfile.TemporaryDirectory() as d:
            os.makedirs(os.path.join(d, "matrix"))
            for t in range(T):
                np.save(os.path.join(d, "matrix", f"{t:04}.npy"), M[t] if t == 0 else M[t].astype(np.float32))
            cm = ref_cm.CoordMap.__new__(ref_cm.CoordMap)
            cm.data_path = d + "/"
            cm.coords, cm.matrices = cm.load_matrix(0, T)            # the reference's loader and quaternion step
        cm.num_coords = cm.coords.shape[1]
        cm.bounding_box = 0.83 + 0.1 * seed
        out[f"{tag}.matrices"] = M
        out[f"{tag}.bounding_box"] = np.float64(cm.bounding_box)
        out[f"{tag}.coords"] = cm.coords
        out[f"{tag}.scale"] = np.float64(cm.get_scale())
        for diff in (True, False):
            cmap, smap = cm.coord_dist_map(diff=diff)
            out[f"{tag}.diff{int(diff)}.map"], out[f"{tag}.diff{int(diff)}.sum"] = cmap, smap
        cmap, smap = cm.coord_dist_map_legacy(diff=False)
        out[f"{tag}.legacy.map"], out[f"{tag}.legacy.sum"] = cmap, smap
    path = os.path.join(HERE, "coord_map_reference.npz")
    np.savez_compressed(path, **out)
    print(f"coord_map_reference.npz {os.path.getsize(path) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
