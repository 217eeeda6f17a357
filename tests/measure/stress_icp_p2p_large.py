"""Randomised parity of point-to-point ICP in K4's many-workgroup regime (round 5) against the oracle (open3d registration_icp restated,
oracle/icp.py; the C search above 4e6 pairs) and against the one-workgroup kernel: k pairs per call (1..4), 1100..30000 sources and
targets per pair, random rigid offsets, noise, thresholds; pose 1e-8, iteration counts equal.
    python tests/measure/stress_icp_p2p_large.py [n_cases] [seed]        (GPU box)"""
import os
import sys

import numpy as np
import torch
from scipy.spatial.transform import Rotation

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from autourdf_amd import ops  # noqa: E402
from oracle import icp as oicp  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
bad = 0
for case in range(n_cases):
    k = int(rng.integers(1, 5))
    ns, nt, srcs, tgts = [], [], [], []
    for i in range(k):
        m = int(rng.integers(1100, 30000 // k + 1100))
        n = int(rng.integers(1100, m + 1))
        shape = rng.uniform(0.05, 1.0, size=3)
        t = rng.uniform(-0.5, 0.5, size=(m, 3)) * shape + 3.0 * i
        if rng.random() < 0.3:
            t = np.round(t * 256) / 256                                # lattice: exact distance ties
        R = Rotation.from_rotvec(rng.normal(scale=0.02, size=3)).as_matrix()
        s = t[rng.permutation(m)[:n]] @ R.T + rng.normal(scale=3e-3, size=3) + rng.normal(scale=rng.choice([0.0, 1e-4, 1e-3]), size=(n, 3))
        ns.append(n); nt.append(m); srcs.append(s); tgts.append(t)
    th = float(rng.choice([0.01, 0.05, 1.0]))
    so = torch.tensor(np.cumsum([0] + ns), dtype=torch.int32, device=dev)
    to = torch.tensor(np.cumsum([0] + nt), dtype=torch.int32, device=dev)
    args = (torch.as_tensor(np.concatenate(srcs), device=dev), so, torch.as_tensor(np.concatenate(tgts), device=dev), to,
            torch.eye(4, dtype=torch.float64, device=dev).repeat(k, 1, 1))
    T, moved, it = ops.icp_p2p(*args, th=th, max_iteration=200)
    os.environ["CREG_ICP_P2P_ONE_WORKGROUP"] = "1"
    T1, m1, it1 = ops.icp_p2p(*args, th=th, max_iteration=200)
    del os.environ["CREG_ICP_P2P_ONE_WORKGROUP"]
    ok = True
    for i in range(k):
        T_ref, _, _, n_ref = oicp.registration_icp(srcs[i], tgts[i], th, np.eye(4), 200)
        e = float(np.abs(T[i].cpu().numpy() - T_ref).max())
        ok = ok and e <= 1e-8 and int(it[i]) == n_ref == int(it1[i])
    ok = ok and float((T - T1).abs().max()) <= 1e-12
    bad += not ok
    print(f"case {case:3d}: k={k} sources {ns} targets {nt} th={th}: iterations {it.tolist()} {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
