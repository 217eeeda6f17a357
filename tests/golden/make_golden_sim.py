"""Mint tests/golden/sim_angle_list.npz (run in the BUILD CONTAINER only).

    python tests/golden/make_golden_sim.py

Source of truth: the reference's own ``angle_list`` (Sim/sim_data.py:372-430) imported with empty stubs for
pybullet / pybullet_data (module-top imports it never touches) and the open3d shim.  Fixture = inputs +
expected outputs only.
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


if __name__ == "__main__":
    main()
