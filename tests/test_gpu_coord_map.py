"""N2: pose-sequence distance maps on the GPU vs the reference's own loops (golden) and
the oracle, through the C ABI (creg_coord_dist_map_f64, creg_pose_coords_f64) and the CoordMap mirror."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_coord_dist_map_vs_reference_golden(dev, golden, tag):
    from autourdf_amd import ops
    g = golden("coord_map_reference.npz")
    M = torch.from_numpy(g[f"{tag}.matrices"]).to(dev)
    bbox = float(g[f"{tag}.bounding_box"])
    np.testing.assert_allclose(ops.pose_coords(M).cpu().numpy(), g[f"{tag}.coords"], atol=1e-12)
    for diff in (True, False):
        d_map, s_map = ops.coord_dist_map(M, bbox, diff)
        # 1e-8 on maps of magnitude ~1: static clusters have relative rotations I + O(1e-8) (float32 files vs the
        # float64 frame 0), whose rotation vectors are pure rounding noise; asin / acos near 0 turn last-bit
        # differences of the 3x3 products into ~3e-9 (measured), in the reference as much as here
        np.testing.assert_allclose(d_map.cpu().numpy(), g[f"{tag}.diff{int(diff)}.map"], atol=1e-8)
        np.testing.assert_allclose(s_map.cpu().numpy(), g[f"{tag}.diff{int(diff)}.sum"], atol=1e-7)


@pytest.mark.parametrize("T,K", [(2, 1), (5, 20), (12, 64), (7, 65), (30, 128)])
def test_coord_dist_map_vs_oracle_sizes(dev, T, K):
    """LDS path (K <= 64), workspace path (K > 64), degenerate K = 1, identity rotations at step 0."""
    from scipy.spatial.transform import Rotation
    from autourdf_amd import ops
    from oracle import coord_map as ocm
    rng = np.random.default_rng(T * 1000 + K)
    M = np.tile(np.eye(4), (T, K, 1, 1))
    M[0, :, :3, 3] = rng.uniform(-0.5, 0.5, size=(K, 3))               # R = I at step 0 (cluster_icp.py:91-95)
    for t in range(1, T):
        dR = Rotation.from_rotvec(rng.normal(scale=0.05, size=(K, 3))).as_matrix()
        M[t, :, :3, :3] = dR @ M[t - 1, :, :3, :3]
        M[t, :, :3, 3] = M[t - 1, :, :3, 3] + rng.normal(scale=0.01, size=(K, 3))
    M[:, K // 2] = M[:, 0]                                             # two identical pose tracks
    for diff in (True, False):
        want_map, want_sum = ocm.coord_dist_map(M, 0.9, diff)
        d_map, s_map = ops.coord_dist_map(torch.from_numpy(M).to(dev), 0.9, diff)
        np.testing.assert_allclose(d_map.cpu().numpy(), want_map, atol=2e-8 if not diff else 1e-9)
        np.testing.assert_allclose(s_map.cpu().numpy(), want_sum, atol=2e-7 if not diff else 1e-8)
        # properties: symmetric, zero diagonal, identical tracks are at distance 0
        d = d_map.cpu().numpy()
        np.testing.assert_allclose(d, d.transpose(1, 0, 2), atol=1e-12)
        assert np.abs(d[np.arange(K), np.arange(K)]).max() < 1e-7
        assert np.abs(d[0, K // 2]).max() < 1e-7


def test_coord_map_class_from_match_layout(dev, golden, tmp_path):
    """The CoordMap mirror on the on-disk layout match() writes (matrix/%04d.npy, cluster/%04d.npz) plus raw PLYs."""
    from autourdf_amd.coord_map import CoordMap
    from autourdf_amd.helper_functions import save_pc_npz
    from oracle import coord_map as ocm
    g = golden("coord_map_reference.npz")
    M = g["a.matrices"]
    T, K = M.shape[:2]
    part, raw = tmp_path / "part" / "0", tmp_path / "raw" / "0"
    os.makedirs(part / "matrix"); os.makedirs(part / "cluster")
    rng = np.random.default_rng(0)
    for t in range(T):
        np.save(part / "matrix" / f"{t:04}.npy", M[t] if t == 0 else M[t].astype(np.float32))
        save_pc_npz([rng.normal(size=(5 + k, 3)) for k in range(K)], str(part / "cluster" / f"{t:04}.npz"))
        os.makedirs(raw / f"{t:04}")
        pts = rng.uniform(-0.4, 0.4, size=(50, 3))
        with open(raw / f"{t:04}" / "robot.ply", "w") as f:
            f.write("ply\nformat ascii 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
            f.write("\n".join(" ".join(f"{v:.9g}" for v in p) for p in pts) + "\n")
    cm = CoordMap(str(part) + "/", str(raw) + "/", start_steps=0, end_steps=T)
    assert cm.num_coords == K and len(cm.clusters) == T and cm.matrices.dtype == np.float64
    np.testing.assert_allclose(cm.coords, g["a.coords"], atol=1e-12)
    assert abs(cm.scale - float(g["a.scale"])) < 1e-12
    assert 0.5 < cm.bounding_box < 2.0
    for diff in (True, False):
        want_map, want_sum = ocm.coord_dist_map(M, cm.bounding_box, diff)
        d_map, s_map = cm.coord_dist_map(diff=diff)
        np.testing.assert_allclose(d_map, want_map, atol=1e-8)
        np.testing.assert_allclose(s_map, want_sum, atol=1e-7)
    lm, ls = cm.coord_dist_map_legacy(diff=False)
    np.testing.assert_allclose(lm, g["a.legacy.map"], atol=1e-12)
    np.testing.assert_allclose(ls, g["a.legacy.sum"], atol=1e-12)
    cm2 = CoordMap.from_arrays(M, cm.bounding_box)
    np.testing.assert_array_equal(cm2.coord_dist_map(True)[0], cm.coord_dist_map(True)[0])
