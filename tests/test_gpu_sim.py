"""N4: synthetic frames from a URDF + meshes: surface sampling on the GPU (creg_sample_mesh_f64)
bit-exact vs the numpy oracle, geometric properties, and the on-disk layout the registration path reads."""
import os

import numpy as np
import pytest
import torch

from _toy_urdf import write_toy_robot

pytestmark = pytest.mark.gpu


def test_sample_mesh_bit_exact_vs_oracle_and_on_surface(tmp_path):
    from autourdf_amd import ops
    from autourdf_amd.sim_data import SimEnv
    from oracle import sim_data as osim
    path, links, joints = write_toy_robot(str(tmp_path))
    env = SimEnv(path, base_position=[0.05, -0.02, 0.0], base_orientation=[0.0, 0.1, 0.7], dof=3)
    q = env.set_joint_positions([0.4, -0.6, 0.9])
    T = env.robot.fk(q, env.base)
    rng = np.random.default_rng(1)
    u = rng.random((20000, 3))
    u[:4] = [[0, 0, 0], [0.999999999, 0.5, 0.5], [0.5, 0, 0.3], [0.25, 1 - 1e-16, 0]]       # table ends, degenerate barycentrics
    dev = torch.device("cuda")
    r = env.robot
    pts, own = ops.sample_mesh(torch.as_tensor(r.tri, device=dev), torch.as_tensor(r.cum_area, device=dev),
                               torch.as_tensor(r.tri_link, device=dev), torch.as_tensor(T, device=dev),
                               torch.as_tensor(u, device=dev), with_links=True)
    want, want_own = osim.sample_mesh(r.tri, r.cum_area, r.tri_link, T, u)
    np.testing.assert_array_equal(own.cpu().numpy(), want_own)
    np.testing.assert_array_equal(pts.cpu().numpy(), want)                  # same operation order: bit for bit
    # every point lies on its link's surface: back in the link frame it is on a face of that link's boxes
    p = pts.cpu().numpy()
    loc = np.einsum("nij,nj->ni", np.linalg.inv(T)[want_own][:, :3, :3], p) + np.linalg.inv(T)[want_own][:, :3, 3]
    base = loc[want_own == 0]
    on_face = (np.isclose(np.abs(base[:, 0]), 0.1, atol=1e-12) | np.isclose(np.abs(base[:, 1]), 0.1, atol=1e-12)
               | np.isclose(base[:, 2], 0.0, atol=1e-12) | np.isclose(base[:, 2], 0.04, atol=1e-12))
    assert on_face.all() and (np.abs(base[:, :2]) <= 0.1 + 1e-12).all()
    tip = loc[want_own == 4]
    assert len(tip) > 0 and (np.linalg.norm(tip, axis=1) <= 0.015 + 1e-12).all()
    # link shares follow the surface areas (20000 samples: 4 sigma)
    area = np.diff(np.concatenate([[0], r.cum_area]))
    for l in range(5):
        share = area[r.tri_link == l].sum() / r.cum_area[-1]
        assert abs((want_own == l).mean() - share) < 4 * np.sqrt(share * (1 - share) / len(u)) + 1e-3


def test_data_collection_writes_the_raw_layout_the_registration_reads(tmp_path):
    from autourdf_amd.cluster_icp import Segments
    from autourdf_amd.sim_data import SimEnv, angle_list, data_collection
    path, _, _ = write_toy_robot(str(tmp_path / "robot"))
    env = SimEnv(path, dof=3)
    a = angle_list(4, 4, 3, env.joint_limits, np.array([0.9] * 3), seed_i=0)
    raw = str(tmp_path / "raw" / "V0000") + "/"
    collision, record = data_collection(env, data_path=raw, angle_list=a, noise_flag=True, num_points=600, seed=3)
    assert collision is False and len(record) == 4 and all(len(c.points) == 600 for c in record)
    assert sorted(os.listdir(raw)) == ["0000", "0001", "0002", "0003", "noise.txt"]
    assert np.loadtxt(raw + "noise.txt").shape == (3, 3)                       # no noise on the first frame
    cfg = open(raw + "0002/joint_cfg.txt").read().split()
    assert [c.split(":")[0] for c in cfg] == ["waist", "shoulder", "wrist"]
    assert abs(float(cfg[1].split(":")[1]) - a[2, 1]) < 1e-6
    seg = Segments(raw)                                                        # the loader of the registration path
    assert seg.data_size == 4
    np.testing.assert_array_equal(np.asarray(seg.pc_list[1].points), record[1].points)
    # deterministic for a seed; the first (noise-free) frame lies exactly on the posed surface
    _, again = data_collection(env, data_path=None, angle_list=a, noise_flag=True, num_points=600, seed=3)
    np.testing.assert_array_equal(again[2].points, record[2].points)
    z0 = record[0].points[:, 2]
    assert z0.min() >= -1e-12 and z0.max() < 0.8
    # farthest-point down-sampling spreads the points: no two of the 600 closer than a regular dense sample would be
    d = np.linalg.norm(record[0].points[:, None] - record[0].points[None], axis=-1) + np.eye(600)
    assert d.min() > 1e-3


def test_collect_layout_and_registration_smoke(tmp_path):
    """collect() -> data/raw/... -> Segments + k-means++ frame-0 segmentation + one ICP-style frame on the result."""
    from autourdf_amd.cluster_icp import Segments
    from autourdf_amd.engine import IcpRegistrar
    from autourdf_amd.sim_data import collect
    write_toy_robot(str(tmp_path / "Robot" / "toy"))
    params = {"gt": "Robot/toy/toy.urdf", "dof": 3, "sim_ori": [0, 0, 0.3]}
    paths = collect("toy", params, num_step=3, step_size=4, epochs=2, num_points=800, root=str(tmp_path))
    assert [os.path.basename(os.path.dirname(p)) for p in paths] == ["V0000", "V0001"]
    assert "data/raw/toy/4_deg_20_cams" in paths[0]
    seg = Segments(paths[0])
    new_pcd = seg.k_means_cluster(pc_id=0, num=6, seed=0)
    assert len(seg.init_matrix_list) == 6 and sum(len(c) for c in seg.init_segment_list) == 800
    reg = IcpRegistrar(np.array(seg.init_matrix_list), seg.init_segment_list, "cuda")
    M, dq, n_it = reg.step(torch.as_tensor(np.asarray(seg.pc_list[1].points), device="cuda"))
    assert torch.isfinite(M).all() and torch.isfinite(dq).all() and (n_it >= 1).all()


def test_camera_visibility_depth_buffers_bit_exact_vs_oracle(tmp_path):
    """creg_visibility_f64: per-camera fp64 depth buffers of the posed triangles and the visibility flags of sampled
    surface points against the numpy restatement (same operation order: identical buffers); and the geometry of it --
    every camera sees something, hidden faces exist, and a point survives exactly when some camera's buffer says so."""
    from autourdf_amd import ops
    from autourdf_amd.sim_data import SimEnv
    from oracle import sim_data as osim
    path, _, _ = write_toy_robot(str(tmp_path))
    env = SimEnv(path, dof=3, radius=1.2, num_cameras=3)
    q = env.set_joint_positions([0.4, -0.6, 0.9])
    rng = np.random.default_rng(2)
    pts = env.sample_surface(q, 6000, rng)
    dev = pts.device
    r = env.robot
    T = env.robot.fk(q, env.base)
    vis, depth = ops.visibility(torch.as_tensor(r.tri, device=dev), torch.as_tensor(r.tri_link, device=dev), torch.as_tensor(T, device=dev),
                                torch.as_tensor(env.cam_frames, device=dev), pts, width=96, height=96, eps=0.004, return_depth=True)
    ovis, odepth = osim.visibility(r.tri, r.tri_link, T, env.cam_frames, pts.cpu().numpy(), width=96, height=96, eps=0.004)
    np.testing.assert_array_equal(depth.cpu().numpy(), odepth)
    np.testing.assert_array_equal(vis.cpu().numpy(), ovis)
    finite = np.isfinite(odepth)
    assert finite.reshape(3, -1).any(1).all() and (odepth[finite] > 0.8).all() and (odepth[finite] < 1.6).all()
    frac = ovis.mean()
    assert 0.5 < frac < 0.98                                        # the base plate's underside and inner faces are hidden
    # the underside of the base box (z = 0 in the base frame) cannot be seen from cameras above the ground plane -- away
    # from its rim, where a point is less than eps behind the side face a camera does see
    ph = pts.cpu().numpy()
    under = np.isclose(ph[:, 2], 0.0, atol=1e-12) & (np.abs(ph[:, 0]) < 0.08) & (np.abs(ph[:, 1]) < 0.08)
    assert under.sum() > 50 and not ovis[under].any()


def test_visibility_through_the_reference_minted_camera_ring(tmp_path, golden):
    """N4 with the part of the reference that can be pinned: the 20-camera ring exactly as the reference's SimEnv._setup_cameras drew
    it (tests/golden/sim_cameras.npz, seeded global RandomState) -- the drop-in builds the same ring, and the depth pass over it equals
    the numpy restatement bit for bit.  What stays unpinned is the PyBullet / OpenGL depth render this pass replaces."""
    from autourdf_amd import ops
    from autourdf_amd.sim_data import SimEnv
    from oracle import sim_data as osim
    g = golden("sim_cameras.npz")
    path, _, _ = write_toy_robot(str(tmp_path))
    radius, n, seed = g["r20.args"]
    np.random.seed(int(seed))
    env = SimEnv(path, dof=3, radius=float(radius), num_cameras=int(n))
    np.testing.assert_array_equal(env.cam_frames[:, :3], g["r20.pos"])
    q = env.set_joint_positions([0.3, 0.5, -0.4])
    pts = env.sample_surface(q, 3000, np.random.default_rng(5))
    dev = pts.device
    r, T = env.robot, env.robot.fk(q, env.base)
    vis, depth = ops.visibility(torch.as_tensor(r.tri, device=dev), torch.as_tensor(r.tri_link, device=dev), torch.as_tensor(T, device=dev),
                                torch.as_tensor(env.cam_frames, device=dev), pts, width=64, height=64, eps=0.004, return_depth=True)
    ovis, odepth = osim.visibility(r.tri, r.tri_link, T, env.cam_frames, pts.cpu().numpy(), width=64, height=64, eps=0.004)
    np.testing.assert_array_equal(depth.cpu().numpy(), odepth)
    np.testing.assert_array_equal(vis.cpu().numpy(), ovis)
    assert 0.3 < ovis.mean() <= 1.0


def test_data_collection_with_occlusion_keeps_only_visible_surface(tmp_path):
    from autourdf_amd.sim_data import SimEnv, angle_list, data_collection
    path, _, _ = write_toy_robot(str(tmp_path / "robot"))
    env = SimEnv(path, dof=3, radius=1.2, num_cameras=4)
    a = angle_list(2, 4, 3, env.joint_limits, np.array([0.9] * 3), seed_i=0)
    _, rec_occ = data_collection(env, angle_list=a, noise_flag=False, num_points=500, seed=1, width=200, height=200)
    _, rec_all = data_collection(env, angle_list=a, noise_flag=False, num_points=500, seed=1, occlusion=False)
    assert all(len(c.points) == 500 for c in rec_occ)
    # no point of the occluded frames lies on the underside of the base plate (away from its rim); the unculled ones do
    inner = lambda c: np.isclose(np.asarray(c.points)[:, 2], 0.0, atol=1e-9) & (np.abs(np.asarray(c.points)[:, :2]) < 0.08).all(1)
    assert not inner(rec_occ[0]).any()
    assert inner(rec_all[0]).any()
