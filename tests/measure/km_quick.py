import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from autourdf_amd import ops
from autourdf_amd.synthetic import initial_segmentation, make_sequence
dev = torch.device("cuda")
N, K = 262144, 128
seq = make_sequence("chain32", 0, 3, N)
mats, _, _ = initial_segmentation(seq[0], K, seed=0, iters=8)
init = torch.as_tensor(mats[:, :3, 3], dtype=torch.float64, device=dev).contiguous()
for f in seq[1:]:
    X = torch.as_tensor(f, dtype=torch.float64, device=dev)
    ops.kmeans_lloyd(X, init)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        _, _, _, n_it = ops.kmeans_lloyd(X, init)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{int(n_it)} Lloyd iterations in {dt*1e3:.2f} ms = {dt*1e6/int(n_it):.1f} us per iteration")
