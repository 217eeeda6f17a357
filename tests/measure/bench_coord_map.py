"""N2 measurement: creg_coord_dist_map_f64 on the GPU vs the oracle (vectorised numpy restatement of
CoordMap.coord_dist_map) on the host, same inputs.  Writes one line per size.

    python tests/measure/bench_coord_map.py > gpurun_out/coord_map.log
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from autourdf_amd import ops          # noqa: E402
from oracle import coord_map as ocm   # noqa: E402  (checker / CPU baseline only)
from scipy.spatial.transform import Rotation  # noqa: E402


def poses(T, K, seed):
    rng = np.random.default_rng(seed)
    M = np.tile(np.eye(4), (T, K, 1, 1))
    M[0, :, :3, 3] = rng.uniform(-0.5, 0.5, size=(K, 3))
    for t in range(1, T):
        M[t, :, :3, :3] = Rotation.from_rotvec(rng.normal(scale=0.05, size=(K, 3))).as_matrix() @ M[t - 1, :, :3, :3]
        M[t, :, :3, 3] = M[t - 1, :, :3, 3] + rng.normal(scale=0.01, size=(K, 3))
    return M


def main():
    dev = torch.device("cuda")
    for T, K in ((10, 20), (10, 40), (50, 30), (200, 128)):
        M = poses(T, K, T + K)
        Md = torch.from_numpy(M).to(dev)
        for diff in (True, False):
            t0 = time.perf_counter()
            want, _ = ocm.coord_dist_map(M, 0.9, diff)
            cpu_ms = (time.perf_counter() - t0) * 1e3
            got, _ = ops.coord_dist_map(Md, 0.9, diff)
            err = float(np.abs(got.cpu().numpy() - want).max())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50
            e0.record()
            for _ in range(reps):
                ops.coord_dist_map(Md, 0.9, diff)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            Tn = T - 1 if diff else T
            alg_bytes = T * K * 128 + K * K * Tn * 8 + K * K * 8
            flops = Tn * (K ** 3) * 6 if diff else Tn * K * K * 30
            print(f"T={T:4d} K={K:4d} diff={int(diff)}  gpu {us:9.1f} us/call (2 launches)  oracle-numpy {cpu_ms:9.2f} ms  "
                  f"x{cpu_ms * 1e3 / us:8.1f}  max|err| {err:.2e}  alg {alg_bytes / 1e3:8.1f} KB -> {alg_bytes / us / 1e3:7.2f} GB/s, "
                  f"{flops / us / 1e6:7.3f} TFLOP/s fp64")


if __name__ == "__main__":
    main()
