"""Debug build only (CREG_EXTRA_FLAGS=-DCREG_ICP_BLK): per-workgroup timeline of the last k_icp_nn launch of a frame that had at most
four clusters still iterating, at the configs[4] shape -- which chunks take how long, and how much they scan."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from autourdf_amd import _lib, ops                                      # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence  # noqa: E402

L = _lib.load()
fn = L.creg_debug_icp_blk
fn.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
N, K = 262144, 128
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seq = make_sequence("chain32", 0, nfr + 1, N)
mats0, clusters0, _ = initial_segmentation(seq[0], K, seed=0, iters=8)
M = torch.as_tensor(mats0, dtype=torch.float64, device=dev).contiguous()
local, off = ops.pack_clusters(clusters0, dev, torch.float64)
out = (ctypes.c_ulonglong * (6 * 8192))()
for t, f in enumerate(seq[1:]):
    f64 = torch.as_tensor(f, dtype=torch.float64, device=dev)
    world32 = ops.cluster_transform(local.to(torch.float32), off, M.to(torch.float32))
    M_new, _, n_it = ops.masked_icp(local, world32, off, f64, M)
    torch.cuda.synchronize()
    fn(out)
    b = np.array(list(out), dtype=np.uint64).reshape(6, 8192)
    grid = int(b[5].max())
    live = (b[1][:grid] > 0) & (b[0][:grid] + np.uint64(40000) > b[0][:grid].max())      # the launches of the last 400 us that still searched
    print(f"frame {t}: grid {grid}, live {int(live.sum())}, iterations max {int(n_it.max())}", flush=True)
    if grid and live.any():
        st = b[0][:grid].astype(np.float64) / 100.0
        en = b[1][:grid].astype(np.float64) / 100.0
        t0 = st.min()
        dur = (en - st)[live]
        per = b[2][:grid][live].astype(np.int64)
        rows = (b[3][:grid][live] >> np.uint64(16)).astype(np.int64)
        cols = (b[3][:grid][live] & np.uint64(0xffff)).astype(np.int64)
        cl = b[4][:grid][live].astype(np.int64)
        sizes = np.diff(off.cpu().numpy())
        o = np.argsort(-dur)
        print(f"frame {t}: ICP iterations max {int(n_it.max())}; last tail launch: grid {grid}, {int(live.sum())} live chunks of clusters "
              f"{sorted(set(cl.tolist()))} (sizes {[int(sizes[c]) for c in sorted(set(cl.tolist()))]}); starts spread {st.max() - t0:.1f} us, "
              f"last search end {en[live].max() - t0:.1f} us")
        print(f"    chunk duration us: min {dur.min():.1f} 50% {np.median(dur):.1f} 90% {np.quantile(dur, 0.9):.1f} max {dur.max():.1f}; entries per lane group (wave 0): "
              f"50% {int(np.median(per))} 90% {int(np.quantile(per, 0.9))} max {per.max()}; corr(duration, entries) {np.corrcoef(dur, per)[0, 1]:.2f}")
        print("    slowest: " + "  ".join(f"[{dur[i]:.0f} us, {per[i]} entries, {rows[i]}x{cols[i]} cells]" for i in o[:8]))
    _, labels, _, _ = ops.kmeans_lloyd(f64, M_new[:, :3, 3].contiguous())
    local, off = ops.group_to_local(f64, labels, M_new)
    M = M_new
