"""Point-to-point ICP of WHOLE clouds (Sim/evaluation.py:358-362 through evaluation.icp_filter): K4's many-workgroup regime (round 5)
beside the one-workgroup kernel that served it before (CREG_ICP_P2P_ONE_WORKGROUP=1), same inputs, same poses.

    python tests/measure/bench_icp_p2p_regimes.py        (GPU box)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from autourdf_amd import ops  # noqa: E402
from scipy.spatial.transform import Rotation  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(21)
for n in (5000, 20000, 60000):
    u, v = rng.uniform(0, 2 * np.pi, n), rng.uniform(0, 2 * np.pi, n)
    gt = np.stack([(0.3 + 0.08 * np.cos(v)) * np.cos(u), (0.3 + 0.08 * np.cos(v)) * np.sin(u), 0.08 * np.sin(v)], 1)
    R = Rotation.from_rotvec([0.004, -0.003, 0.005]).as_matrix()
    pred = gt[rng.permutation(n)] @ R.T + np.array([0.0015, -0.001, 0.002]) + rng.normal(scale=2e-4, size=(n, 3))
    off = torch.tensor([0, n], dtype=torch.int32, device=dev)
    args = (torch.as_tensor(pred, device=dev), off, torch.as_tensor(gt, device=dev), off, torch.eye(4, dtype=torch.float64, device=dev)[None])
    res = {}
    for regime in ("many workgroups", "one workgroup"):
        if regime == "one workgroup":
            os.environ["CREG_ICP_P2P_ONE_WORKGROUP"] = "1"
        else:
            os.environ.pop("CREG_ICP_P2P_ONE_WORKGROUP", None)
        ops.icp_p2p(*args, th=0.01, max_iteration=20000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T, moved, it = ops.icp_p2p(*args, th=0.01, max_iteration=20000)
        torch.cuda.synchronize()
        res[regime] = (time.perf_counter() - t0, T.cpu().numpy(), int(it[0]))
    a, b = res["many workgroups"], res["one workgroup"]
    print(f"icp_filter {n} x {n}: many workgroups {a[0] * 1e3:9.2f} ms ({a[2]} iterations)   one workgroup {b[0] * 1e3:9.2f} ms ({b[2]} iterations)   "
          f"max |pose difference| {np.abs(a[1] - b[1]).max():.1e}", flush=True)
