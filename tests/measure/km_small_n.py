"""creg_kmeans_lloyd_f64 at small frames, by path (env CREG_KM_PRUNE / CREG_KM_PERSIST): ms per call."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from autourdf_amd import ops
from autourdf_amd.synthetic import initial_segmentation, make_sequence
dev = torch.device("cuda")
for robot, N, K in (("wx200_5", 4096, 20), ("franka", 16384, 40), ("chain32", 32768, 128), ("chain32", 65536, 128)):
    seq = make_sequence(robot, 0, 2, N)
    mats, _, _ = initial_segmentation(seq[0], K, seed=0, iters=8)
    init = torch.as_tensor(mats[:, :3, 3], dtype=torch.float64, device=dev).contiguous()
    X = torch.as_tensor(seq[1], dtype=torch.float64, device=dev)
    ops.kmeans_lloyd(X, init); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        _, _, _, n_it = ops.kmeans_lloyd(X, init)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"N={N} K={K}: {int(n_it)} iterations, {dt*1e3:.3f} ms per call = {dt*1e6/int(n_it):.1f} us per iteration")
