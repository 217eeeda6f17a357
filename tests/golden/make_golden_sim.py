"""Mint tests/golden/sim_angle_list.npz and tests/golden/sim_cameras.npz (run in the BUILD CONTAINER only).

    python tests/golden/make_golden_sim.py

Source of truth: the reference's own ``angle_list`` (Sim/sim_data.py:372-430) imported with empty stubs for
pybullet / pybullet_data (module-top imports it never touches) and the open3d shim.  Fixture = inputs +
expected outputs only.  sim_cameras.npz: the reference's own ``SimEnv._setup_cameras`` (Sim/sim_data.py:85-116), called unbound on
an empty object (it only writes ``self.cameras``) for rings of fewer than 20 cameras (evenly spaced, 20 degrees elevation) and of
20 or more (drawn from numpy's global RandomState, seeded here) -- the part of the reference's data generation that IS plain numpy;
the depth rendering behind it (PyBullet + OpenGL) is not.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
sys.modules["pybullet"] = types.ModuleType("pybullet")
sys.modules["pybullet_data"] = types.ModuleType("pybullet_data")
sys.path.insert(0, "/root/reference/Sim")
import sim_data as ref_sim  # noqa: E402  (reference)


def main():
    out = {}
    cases = {"wx200": (10, 4, 5, np.array([[-3.14, 3.14], [-1.88, 1.97], [-1.88, 1.62], [-1.74, 2.14], [-3.14, 3.14]]), 0.9, 0),
             "franka": (25, 4, 7, np.array([[-2.9, 2.9], [-1.76, 1.76], [-2.9, 2.9], [-3.07, -0.07], [-2.9, 2.9], [-0.02, 3.75], [-2.9, 2.9]]), 0.9, 3),
             "coarse": (6, 10, 2, np.array([[-1.0, 1.0], [0.5, -0.5]]), 0.5, 11)}
    for tag, (num_step, step_size, dof, limits, scale, seed) in cases.items():
        a = ref_sim.angle_list(num_step, step_size, dof, limits.copy(), np.array([scale] * dof), seed)
        out[f"{tag}.args"] = np.array([num_step, step_size, dof, scale, seed], np.float64)
        out[f"{tag}.limits"] = limits
        out[f"{tag}.angles"] = a
    path = os.path.join(HERE, "sim_angle_list.npz")
    np.savez_compressed(path, **out)
    print(f"sim_angle_list.npz {os.path.getsize(path) / 1024:.1f} KB")
    cams = {}
    for tag, (radius, n, seed) in {"r3": (1.5, 3, 0), "r8": (1.0, 8, 0), "r19": (2.0, 19, 0), "r20": (2.5, 20, 4), "r24": (1.2, 24, 2024)}.items():
        holder = types.SimpleNamespace()
        np.random.seed(seed)
        ref_sim.SimEnv._setup_cameras(holder, radius, n)
        c = holder.cameras
        cams[f"{tag}.args"] = np.array([radius, n, seed], np.float64)
        cams[f"{tag}.pos"] = np.array([x["camera_pos"] for x in c], np.float64)
        cams[f"{tag}.target"] = np.array([x["target_pos"] for x in c], np.float64)
        cams[f"{tag}.up"] = np.array([x["up_vector"] for x in c], np.float64)
        cams[f"{tag}.intrinsics"] = np.array([[x["fov"], x["aspect"], x["near_val"], x["far_val"]] for x in c], np.float64)
    path = os.path.join(HERE, "sim_cameras.npz")
    np.savez_compressed(path, **cams)
    print(f"sim_cameras.npz {os.path.getsize(path) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
