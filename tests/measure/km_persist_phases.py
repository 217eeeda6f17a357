"""Debug build only (CREG_EXTRA_FLAGS=-DCREG_STAMPS): where an iteration of the persistent Lloyd kernel spends its time, averaged over
the workgroups and iterations of one k_means() call at the configs[4] shape."""
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
dev = torch.device("cuda")
N, K = 262144, 128
seq = make_sequence("chain32", 0, 3, N)
mats, _, _ = initial_segmentation(seq[0], K, seed=0, iters=8)
init = torch.as_tensor(mats[:, :3, 3], dtype=torch.float64, device=dev).contiguous()
X = torch.as_tensor(seq[2], dtype=torch.float64, device=dev)
ops.kmeans_lloyd(X, init)
torch.cuda.synchronize(); fn(None, 1)
t0 = time.perf_counter()
_, _, _, n_it = ops.kmeans_lloyd(X, init)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out = (ctypes.c_ulonglong * 16)()
fn(out, 4)
v = np.array(list(out), dtype=np.float64)
wi = max(v[15], 1.0)
names = ("centre rows", "pruned sweep", "moves + flush", "arrival", "M-step tail (its workgroup only, per iteration)", "wait for gen")
print(f"{int(n_it)} Lloyd iterations in {dt * 1e3:.2f} ms = {dt * 1e6 / int(n_it):.1f} us per iteration ({int(wi)} workgroup-iterations)")
for p, nm in enumerate(names):
    d = int(n_it) if p == 4 else wi
    print(f"    {nm:48s} {v[p] / d / 100:6.2f} us")
print("  workgroup 0 (runs the tails):")
for p, nm in enumerate(names):
    print(f"    {nm:48s} {v[8 + p] / int(n_it) / 100:6.2f} us")
big = (ctypes.c_ulonglong * 4096)()
fn(big, 5)
b = np.array(list(big), dtype=np.float64).reshape(4, 1024)[:, :768] / 100.0
t0 = b[0, 0]
print("  iteration 150, absolute times relative to workgroup 0's start of the iteration (us):")
for name, row in zip(("centre rows landed", "arrived (wg 0: all seen)", "tail done (wg 0 only)", "saw gen"), b):
    r = np.sort(row[1:768] - t0)
    print(f"    {name:28s}: wg0 {row[0] - t0:6.2f} | others min {r[0]:6.2f}  25% {r[len(r)//4]:6.2f}  50% {r[len(r)//2]:6.2f}  75% {r[3*len(r)//4]:6.2f}  max {r[-1]:6.2f}")
raw = np.array(list(big), dtype=np.uint64).reshape(4, 1024)[:, :768]
arr = raw[1].astype(np.float64) / 100.0 - t0
order = np.argsort(-arr[1:])[:12] + 1
print("  latest arrivals: " + "  ".join(f"wg {i}: {arr[i]:.1f} us, survivors {int(raw[2][i] >> np.uint64(32))}, changed(w0) {int(raw[2][i] & np.uint64(0xffffffff))}" for i in order))
nc = (raw[2][1:] >> np.uint64(32)).astype(int)
print(f"  survivors per workgroup: median {int(np.median(nc))}, 90% {int(np.quantile(nc, 0.9))}, max {nc.max()}")
