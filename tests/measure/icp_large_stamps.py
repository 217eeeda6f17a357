"""Debug build only (CREG_EXTRA_FLAGS=-DCREG_STAMPS): how many targets a wave of k_icp_nn scans at the configs[4] shape."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from autourdf_amd import _lib, ops                                      # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence  # noqa: E402

L = _lib.load()
fn = L.creg_debug_icp_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
fw = L.creg_debug_icp_wall
fw.argtypes = [ctypes.c_void_p, ctypes.c_int]
wout = (ctypes.c_ulonglong * 8)()
dev = torch.device("cuda")
N, K = 262144, 128
seq = make_sequence("chain32", 0, 14, N)
mats0, clusters0, _ = initial_segmentation(seq[0], K, seed=0, iters=8)
M = torch.as_tensor(mats0, dtype=torch.float64, device=dev).contiguous()
local, off = ops.pack_clusters(clusters0, dev, torch.float64)
out = (ctypes.c_ulonglong * (512 * 16))()
for f in seq[1:]:
    f64 = torch.as_tensor(f, dtype=torch.float64, device=dev)
    world32 = ops.cluster_transform(local.to(torch.float32), off, M.to(torch.float32))
    torch.cuda.synchronize(); fn(None, 1); fw(None, 1)
    M_new, _, n_it = ops.masked_icp(local, world32, off, f64, M)
    torch.cuda.synchronize(); fn(out, 0)
    v = np.array(list(out)[:16], dtype=np.float64)
    w = max(v[1], 1)
    print(f"waves {v[1]:.0f}: scan steps per wave {v[0] / w:.1f} (row groups {v[4] / w:.2f}, rows {v[5] / w:.2f}, columns {v[6] / w:.2f}) of {v[3] / w:.0f} masked targets, "
          f"scan cycles per wave {v[7] / w:.0f}, waves with a tie rescan {v[2]:.0f}, iterations mean {n_it.double().mean():.1f} max {int(n_it.max())}")
    fw(wout, 0); wv_ = np.array(list(wout), dtype=np.float64); nl = max(wv_[3], 1)
    print('   tail launches (<= 4 clusters iterating): %d; from the launch start (first block): first live block starts after %.1f us, last search ends after %.1f us, fit ends after %.1f us' % (wv_[3], wv_[0] / nl / 100, wv_[1] / nl / 100, wv_[2] / nl / 100))
    nb = max(v[15], 1)
    print('   wave 0 of a block, cycles per block: load+update %.0f, bounds+rows %.0f, staging %.0f, scan %.0f, combine %.0f, moments %.0f, barrier wait %.0f  (%d blocks)' % (v[8] / nb, v[9] / nb, v[10] / nb, v[11] / nb, v[12] / nb, v[13] / nb, v[14] / nb, nb))
    _, labels, _, _ = ops.kmeans_lloyd(f64, M_new[:, :3, 3].contiguous())
    local, off = ops.group_to_local(f64, labels, M_new)
    M = M_new
