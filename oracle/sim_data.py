"""Oracle for the synthetic-frame generator.  TEST INFRASTRUCTURE ONLY.

* ``angle_list``: the reference's own function (Sim/sim_data.py:372-430) is imported under shims by
  tests/golden/make_golden_sim.py -> tests/golden/sim_angle_list.npz                     -- PINNED
* ``sample_mesh``: numpy restatement of creg_sample_mesh_f64 with the same operation order (bit-exact check)
* ``fk``: an independent forward-kinematics evaluation (scipy Rotation) for the URDF subset the generator
  reads.  The reference delegates both to PyBullet + OpenGL rendering (not restatable)   -- parity UNPINNED
"""
import numpy as np
from scipy.spatial.transform import Rotation


def sample_mesh(tri, cum_area, tri_link, link_T, u):
    tri, cum_area, link_T, u = (np.asarray(a, np.float64) for a in (tri, cum_area, link_T, u))
    target = u[:, 0] * cum_area[-1]
    f = np.minimum(np.searchsorted(cum_area, target, side="right"), len(cum_area) - 1)
    s = np.sqrt(u[:, 1])
    b0, b1, b2 = 1.0 - s, s * (1.0 - u[:, 2]), s * u[:, 2]
    t = tri[f]
    p = (b0[:, None] * t[:, 0] + b1[:, None] * t[:, 1]) + b2[:, None] * t[:, 2]
    T = link_T[np.asarray(tri_link)[f]]
    out = np.empty_like(p)
    for d in range(3):
        out[:, d] = ((T[:, d, 0] * p[:, 0] + T[:, d, 1] * p[:, 1]) + T[:, d, 2] * p[:, 2]) + T[:, d, 3]
    return out, np.asarray(tri_link)[f]


def fk(links, joints, q, root, base=None):
    """links: names; joints: dicts with name/type/parent/child/xyz/rpy/axis; q: {name: value}."""
    T = {root: np.eye(4) if base is None else np.asarray(base, np.float64)}
    todo = list(joints)
    while todo:
        j = next(j for j in todo if j["parent"] in T)
        todo.remove(j)
        O = np.eye(4)
        O[:3, :3] = Rotation.from_euler("xyz", j["rpy"]).as_matrix()          # extrinsic xyz = URDF fixed-axis rpy
        O[:3, 3] = j["xyz"]
        M = np.eye(4)
        a = np.asarray(j["axis"], np.float64)
        a = a / np.linalg.norm(a)
        v = float(q.get(j["name"], 0.0))
        if j["type"] in ("revolute", "continuous"):
            M[:3, :3] = Rotation.from_rotvec(a * v).as_matrix()
        elif j["type"] == "prismatic":
            M[:3, 3] = a * v
        T[j["child"]] = T[j["parent"]] @ O @ M
    return np.stack([T[l] for l in links])
