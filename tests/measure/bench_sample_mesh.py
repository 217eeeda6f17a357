"""N4 measurement: creg_sample_mesh_f64 (area-weighted surface sampling of a posed mesh) on the GPU vs the numpy
oracle, same inputs; algorithmic bytes = 24 (uniforms) + 72 (triangle) + 24 (point) per sample.

    python tests/measure/bench_sample_mesh.py > gpurun_out/sample_mesh.log
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from autourdf_amd import ops            # noqa: E402
from oracle import sim_data as osim     # noqa: E402  (checker / CPU baseline only)


def main():
    dev = torch.device("cuda")
    rng = np.random.default_rng(0)
    for F, L, n in ((20_000, 8, 20_000), (200_000, 16, 1 << 20), (200_000, 16, 1 << 24)):
        tri = rng.normal(size=(F, 3, 3)) * 0.01 + rng.normal(size=(F, 1, 3)) * 0.2
        area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
        cum = np.cumsum(area)
        own = rng.integers(0, L, size=F).astype(np.int32)
        T = np.tile(np.eye(4), (L, 1, 1)); T[:, :3, 3] = rng.normal(size=(L, 3))
        u = rng.random((n, 3))
        d = [torch.as_tensor(a, device=dev) for a in (tri, cum, own, T, u)]
        got = ops.sample_mesh(*d)
        m = min(n, 1 << 20)
        t0 = time.perf_counter()
        want, _ = osim.sample_mesh(tri, cum, own, T, u[:m])
        cpu_ms = (time.perf_counter() - t0) * 1e3 * n / m
        exact = bool(np.array_equal(got[:m].cpu().numpy(), want))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ops.sample_mesh(*d)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        alg = 120.0 * n
        print(f"F={F:7d} n={n:9d}  gpu {us:9.1f} us  {alg / us / 1e3:8.1f} GB/s algorithmic ({alg / us / 1e3 / 8000:.3f} of 8 TB/s)  "
              f"oracle-numpy {cpu_ms:9.1f} ms  x{cpu_ms * 1e3 / us:7.0f}  bit-exact {exact}")


if __name__ == "__main__":
    main()
