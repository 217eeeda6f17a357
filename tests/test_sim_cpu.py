"""N4 host logic (no GPU): joint trajectories vs the reference's own function, URDF / mesh readers, FK."""
import os

import numpy as np
import pytest

from _toy_urdf import write_toy_robot


@pytest.mark.parametrize("tag", ["wx200", "franka", "coarse"])
def test_angle_list_equals_reference_golden(golden, tag):
    from autourdf_amd.sim_data import angle_list
    g = golden("sim_angle_list.npz")
    num_step, step_size, dof, scale, seed = g[f"{tag}.args"]
    got = angle_list(int(num_step), int(step_size), int(dof), g[f"{tag}.limits"].copy(), np.array([scale] * int(dof)), int(seed))
    np.testing.assert_array_equal(got, g[f"{tag}.angles"])            # same generator calls, same arithmetic


def test_urdf_meshes_and_fk_vs_independent_oracle(tmp_path):
    from autourdf_amd.sim_data import SimEnv, UrdfRobot, load_mesh
    from oracle import sim_data as osim
    path, links, joints = write_toy_robot(str(tmp_path))
    rob = UrdfRobot(path)
    assert rob.links == links and rob.root == "base" and [j["name"] for j in rob.joints] == [j["name"] for j in joints]
    for f, n in (("l1.stl", 12), ("l2.STL", 12), ("l3.obj", 12)):
        assert load_mesh(os.path.join(str(tmp_path), "meshes", f)).shape == (n, 3, 3)
    counts = np.bincount(rob.tri_link, minlength=5)
    assert list(counts[:4]) == [12, 12, 12, 12] and counts[4] > 100             # sphere primitive
    # surface areas: base box 2(ab+bc+ca); the l2 visual is scaled by 1.2 along z
    area = np.diff(np.concatenate([[0], rob.cum_area]))
    assert abs(area[rob.tri_link == 0].sum() - 2 * (0.2 * 0.2 + 2 * 0.2 * 0.04)) < 1e-12
    assert abs(area[rob.tri_link == 2].sum() - 2 * (0.03 * 0.05 + (0.03 + 0.05) * 0.18)) < 1e-9
    rng = np.random.default_rng(0)
    for _ in range(5):
        q = {j["name"]: float(rng.uniform(-1, 1)) * (0.04 if j["type"] == "prismatic" else 1.0) for j in joints}
        np.testing.assert_allclose(rob.fk(q), osim.fk(links, joints, q, "base"), atol=1e-13)
    env = SimEnv(path, base_position=[0.1, 0, 0], base_orientation=[0, 0, 0.5], dof=2)
    assert env.joint_list == ["waist", "shoulder", "wrist"] and env.dof_list == ["waist", "shoulder"]   # revolute only, URDF order
    np.testing.assert_allclose(env.joint_limits, [[-3.1, 3.1], [-1.5, 1.2]])
    q = env.set_joint_positions([0.3, -0.2, 99.0])
    assert q == {"waist": 0.3, "shoulder": -0.2, "wrist": 0.5}                   # undriven joint parked at mid range
    with pytest.raises(NotImplementedError):
        load_mesh("x.ply")


@pytest.mark.skipif(not os.path.isdir("/root/reference/Robot"), reason="reference assets only exist in the build container")
@pytest.mark.parametrize("rel,dof", [("interbotix_descriptions/urdf/wx200_real.urdf", 5),
                                     ("allegro_hand_description/allegro_hand_description_left.urdf", 16)])
def test_reference_urdfs_parse_when_present(rel, dof):
    """The real assets the generator is for (parameters.json 'gt' entries); read-only, build container only."""
    from autourdf_amd.sim_data import SimEnv
    env = SimEnv(os.path.join("/root/reference/Robot", rel), dof=dof)
    assert len(env.dof_list) == dof and len(env.robot.tri) > 1000
    T = env.robot.fk(env.set_joint_positions(np.zeros(len(env.joint_list))))
    assert np.isfinite(T).all() and np.allclose(np.linalg.det(T[:, :3, :3]), 1.0)


DAE = """<?xml version="1.0" encoding="utf-8"?>
<COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">
  <asset><unit name="centimeter" meter="0.01"/><up_axis>Y_UP</up_axis></asset>
  <library_geometries>
    <geometry id="quad"><mesh>
      <source id="quad-pos"><float_array id="quad-pos-a" count="12">0 0 0  10 0 0  10 20 0  0 20 0</float_array>
        <technique_common><accessor source="#quad-pos-a" count="4" stride="3"/></technique_common></source>
      <source id="quad-nrm"><float_array id="quad-nrm-a" count="3">0 0 1</float_array>
        <technique_common><accessor source="#quad-nrm-a" count="1" stride="3"/></technique_common></source>
      <vertices id="quad-vtx"><input semantic="POSITION" source="#quad-pos"/></vertices>
      <polylist count="1"><input semantic="VERTEX" source="#quad-vtx" offset="0"/><input semantic="NORMAL" source="#quad-nrm" offset="1"/>
        <vcount>4</vcount><p>0 0 1 0 2 0 3 0</p></polylist>
    </mesh></geometry>
    <geometry id="tri"><mesh>
      <source id="tri-pos"><float_array id="tri-pos-a" count="9">0 0 0  1 0 0  0 1 0</float_array>
        <technique_common><accessor source="#tri-pos-a" count="3" stride="3"/></technique_common></source>
      <vertices id="tri-vtx"><input semantic="POSITION" source="#tri-pos"/></vertices>
      <triangles count="1"><input semantic="VERTEX" source="#tri-vtx" offset="0"/><p>0 1 2</p></triangles>
    </mesh></geometry>
  </library_geometries>
  <library_visual_scenes><visual_scene id="s">
    <node id="a"><translate>0 0 5</translate><instance_geometry url="#quad"/>
      <node id="b"><rotate>0 0 1 90</rotate><scale>2 2 2</scale><instance_geometry url="#tri"/></node></node>
  </visual_scene></library_visual_scenes>
  <scene><instance_visual_scene url="#s"/></scene>
</COLLADA>
"""


def test_collada_loader_scene_graph_units_and_up_axis(tmp_path):
    """COLLADA visuals (franka's): polylist + triangles, a nested node with translate / rotate / scale, centimetre units and
    Y_UP converted the way PyBullet's URDF importer does ((x, y, z) -> (x, -z, y))."""
    from autourdf_amd.sim_data import load_mesh
    p = tmp_path / "m.dae"
    p.write_text(DAE)
    t = load_mesh(str(p))
    assert t.shape == (3, 3, 3)
    quad = np.array([[0, 0, 5], [10, 0, 5], [10, 20, 5], [0, 20, 5]], float) * 0.01          # translated, in metres, still Y_UP
    zup = lambda v: np.stack([v[:, 0], -v[:, 2], v[:, 1]], 1)
    np.testing.assert_allclose(t[0], zup(quad[[0, 1, 2]]), atol=1e-15)
    np.testing.assert_allclose(t[1], zup(quad[[0, 2, 3]]), atol=1e-15)
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], float) * 2                                 # scale, then rotate 90 about z, then translate
    tri = np.stack([-tri[:, 1], tri[:, 0], tri[:, 2]], 1) + [0, 0, 5]
    np.testing.assert_allclose(t[2], zup(tri * 0.01), atol=1e-15)


def test_camera_ring_matches_the_reference_formula(tmp_path):
    from autourdf_amd.sim_data import SimEnv
    from oracle import sim_data as osim
    path, _, _ = write_toy_robot(str(tmp_path))
    env = SimEnv(path, dof=2, radius=1.5, num_cameras=3)
    np.testing.assert_allclose(env.cam_frames, osim.camera_ring(1.5, 3), atol=1e-15)
    assert len(env.cameras) == 3 and env.cameras[0]["fov"] == 60 and env.cameras[0]["far_val"] == 4
    for c in env.cam_frames:                                       # orthonormal frames looking at the origin
        f, s, u = c[3:6], c[6:9], c[9:12]
        np.testing.assert_allclose([f @ s, f @ u, s @ u, f @ f, s @ s, u @ u], [0, 0, 0, 1, 1, 1], atol=1e-14)
        np.testing.assert_allclose(np.cross(c[:3] / np.linalg.norm(c[:3]), f), 0, atol=1e-14)
    np.random.seed(4)
    env20 = SimEnv(path, dof=2, radius=2.5, num_cameras=20)
    np.random.seed(4)
    th, ph = np.random.rand(20) * 2 * np.pi, np.random.rand(20) * np.pi / 2
    np.testing.assert_allclose(env20.cam_frames[:, :3], np.stack([2.5 * np.cos(th) * np.cos(ph), 2.5 * np.sin(th) * np.cos(ph), 2.5 * np.sin(ph)], 1))


def test_camera_ring_vs_reference_minted_golden(tmp_path, golden):
    """tests/golden/sim_cameras.npz holds what the REFERENCE's own SimEnv._setup_cameras (Sim/sim_data.py:85-116) produced for rings
    of 3 / 8 / 19 cameras and -- under numpy's seeded global RandomState -- of 20 / 24: the drop-in's camera dictionaries and the
    oracle's ring (whose frames the visibility kernel is checked against) reproduce positions, targets, up vectors and intrinsics."""
    from autourdf_amd.sim_data import SimEnv
    from oracle import sim_data as osim
    path, _, _ = write_toy_robot(str(tmp_path))
    g = golden("sim_cameras.npz")
    for tag in ("r3", "r8", "r19", "r20", "r24"):
        radius, n, seed = g[f"{tag}.args"]
        n, seed = int(n), int(seed)
        np.random.seed(seed)
        env = SimEnv(path, dof=2, radius=float(radius), num_cameras=n)
        pos = np.array([c["camera_pos"] for c in env.cameras])
        np.testing.assert_array_equal(pos, g[f"{tag}.pos"])                      # the same numpy expressions on the same draws
        np.testing.assert_array_equal(np.array([c["target_pos"] for c in env.cameras], np.float64), g[f"{tag}.target"])
        np.testing.assert_array_equal(np.array([c["up_vector"] for c in env.cameras], np.float64), g[f"{tag}.up"])
        np.testing.assert_array_equal(np.array([[c["fov"], c["aspect"], c["near_val"], c["far_val"]] for c in env.cameras], np.float64),
                                      g[f"{tag}.intrinsics"])
        np.testing.assert_array_equal(env.cam_frames[:, :3], g[f"{tag}.pos"])
        # the oracle's ring: its own Generator for >= 20 cameras, so hand it the reference's draws through a RandomState adapter
        class _Adapter:
            def __init__(self, s): self.r = np.random.RandomState(s)
            def random(self, k): return self.r.rand(k)
        ring = osim.camera_ring(float(radius), n, rng=_Adapter(seed))
        np.testing.assert_allclose(ring[:, :3], g[f"{tag}.pos"], atol=1e-15)
        np.testing.assert_allclose(ring, env.cam_frames, atol=1e-15)


@pytest.mark.skipif(not os.path.isdir("/root/reference/Robot"), reason="reference assets only exist in the build container")
def test_reference_franka_collada_visuals_parse_when_present():
    from autourdf_amd.sim_data import SimEnv
    env = SimEnv("/root/reference/Robot/franka/franka_panda.urdf", dof=6)
    assert len(env.robot.tri) > 100000 and len(env.dof_list) == 6
    T = env.robot.fk(env.set_joint_positions(np.zeros(len(env.joint_list))))
    pts = np.concatenate([env.robot.tri[env.robot.tri_link == l].reshape(-1, 3) @ T[l, :3, :3].T + T[l, :3, 3]
                          for l in range(len(T)) if (env.robot.tri_link == l).any()])
    assert 0.9 < pts[:, 2].max() < 1.3 and abs(pts[:, 2].min()) < 0.02        # a Panda standing upright on its base
