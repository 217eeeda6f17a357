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
        load_mesh("x.dae")


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
