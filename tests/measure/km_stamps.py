"""Debug build only (CREG_EXTRA_FLAGS=-DCREG_STAMPS): the timeline of one Lloyd launch at the configs[4] shape."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from autourdf_amd import _lib, ops                                      # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence  # noqa: E402

L = _lib.load()
fn = L.creg_debug_km_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
dev = torch.device("cuda")
N, K = 262144, 128
seq = make_sequence("chain32", 0, 3, N)
mats, _, _ = initial_segmentation(seq[0], K, seed=0, iters=8)
init = torch.as_tensor(mats[:, :3, 3], dtype=torch.float64, device=dev).contiguous()
for f in seq[1:]:
    X = torch.as_tensor(f, dtype=torch.float64, device=dev)
    ops.kmeans_lloyd(X, init)
    torch.cuda.synchronize(); fn(None, 1)
    t0 = time.perf_counter()
    _, _, _, n_it = ops.kmeans_lloyd(X, init)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fn(out, 0)
    v = np.array(list(out), dtype=np.float64)
    nl = max(v[0], 1)
    print(f"{int(n_it)} Lloyd iterations in {dt * 1e3:.2f} ms = {dt * 1e6 / int(n_it):.1f} us per iteration; per launch, from its first block's start: "
          f"last block ends its E-step after {v[5] / nl / 100:.1f} us, last arrival {v[2] / nl / 100:.1f} us, M-step tail done {v[3] / nl / 100:.1f} us ({int(v[0])} launches)")
    fn(out, 2)
    ph = np.array(list(out), dtype=np.float64) / nl / 100
    print(f"    pruned E-step, workgroup 300 past: flags {ph[0]:.1f} us, operand loads {ph[1]:.1f} us, box bounds + survivors {ph[2]:.1f} us, evaluation {ph[3]:.1f} us; "
          f"[issued {ph[4]:.1f}, first X {ph[5]:.1f}, X+prev {ph[6]:.1f}, B {ph[7]:.1f}]")

    big = (ctypes.c_ulonglong * 4096)()
    fn(big, 3)
    b = np.array(list(big), dtype=np.float64).reshape(4, 1024)[:, :512] / 100.0
    t0 = b[0].min()
    for name, row in zip(("start", "operands landed", "E-step end", "after arrival"), b):
        r = np.sort(row - t0)
        print(f"    last launch, per workgroup, {name:16s}: min {r[0]:5.1f}  25% {r[128]:5.1f}  50% {r[256]:5.1f}  75% {r[384]:5.1f}  max {r[-1]:5.1f} us")
    order = np.argsort(b[0]); print("    start order (block ids, first 16 / last 16):", order[:16], order[-16:])
