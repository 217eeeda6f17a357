"""N3 measurement: link refinement (point-to-point ICP of every link of every time step to the first
step) on the GPU vs the oracle's ICP on the host.  wx200_5-shaped: 10 steps x 6 links, N=4096.

    python tests/measure/bench_link_refine.py > gpurun_out/link_refine.log
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from autourdf_amd import ops                      # noqa: E402
from autourdf_amd.synthetic import make_sequence  # noqa: E402
from oracle import link as olink                  # noqa: E402  (checker / CPU baseline only)
from scipy.spatial.transform import Rotation      # noqa: E402


def main():
    dev = torch.device("cuda")
    rng = np.random.default_rng(0)
    T, L, N = 10, 6, 4096
    base = make_sequence("wx200_5", 0, 1, N)[0]
    order = np.argsort(base[:, 2])
    cuts = np.linspace(0, N, L + 1).astype(int)
    first = [base[order[cuts[i]:cuts[i + 1]]] for i in range(L)]
    steps = [first]
    for t in range(1, T):
        cl = []
        for f in first:
            R = Rotation.from_rotvec(rng.normal(scale=0.03, size=3)).as_matrix()
            cl.append(f[rng.permutation(len(f))] @ R.T + rng.normal(scale=0.004, size=3) + rng.normal(scale=5e-4, size=f.shape))
        steps.append(cl)
    cat = lambda cs: torch.as_tensor(np.concatenate(cs), device=dev)
    off = lambda cs: torch.tensor(np.concatenate([[0], np.cumsum([len(c) for c in cs])]), dtype=torch.int32, device=dev)
    init = torch.eye(4, dtype=torch.float64, device=dev).repeat(L, 1, 1)
    probs = [(cat(s), off(s), cat(first), off(first), init) for s in steps]
    outs = ops.icp_p2p_batch(probs, th=1.0, max_iteration=100000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        outs = ops.icp_p2p_batch(probs, th=1.0, max_iteration=100000)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) * 1e3 / reps
    t0 = time.perf_counter()
    ref = olink.refine_links(steps[:3], first, L - 1)
    cpu_ms = (time.perf_counter() - t0) * 1e3 / 3 * T
    err = max(float(np.abs(outs[t][1].cpu().numpy() - np.concatenate(ref[t])).max()) for t in range(3))
    iters = torch.stack([o[2] for o in outs]).double()
    print(f"link refine T={T} links={L} N={N}: gpu {gpu_ms:.2f} ms (one launch, {T * L} workgroups, mean {iters.mean():.1f} / max "
          f"{int(iters.max())} ICP iterations)  oracle-numpy {cpu_ms:.0f} ms (3 steps timed, scaled to {T})  x{cpu_ms / gpu_ms:.0f}  "
          f"max|moved - oracle| {err:.2e}")


if __name__ == "__main__":
    main()
