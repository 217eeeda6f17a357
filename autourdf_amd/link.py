"""Drop-in for ``refine_links_clusters`` of the reference's ``PointCloud/link.py`` (:85-127; SURVEY 8(f)
N3): every link cloud of every time step is registered to the same link at ``start_steps`` by
point-to-point ICP (threshold 1, identity start, open3d's relative 1e-6 stopping rule) and written,
moved, to ``cluster_rf/{t:04}.npz``.  The reference runs one Open3D ICP per (time step, link); here the
links of up to 16 time steps share ONE launch of the K4 kernel in its point-to-point mode
(``creg_masked_icp_batch_f64`` with ``tgt_offsets``; a workgroup per link per time step).

Meshing, SDF and GUI functions of the reference file are out of scope.  No CPU fallback.
"""
import glob
import os

import numpy as np
import torch

from . import _lib, ops
from .helper_functions import load_pc_npz, save_pc_npz


def _pack(clouds, device):
    sizes = [len(c) for c in clouds]
    off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=device)
    pts = torch.as_tensor(np.concatenate([np.asarray(c, np.float64).reshape(-1, 3) for c in clouds]), device=device)
    return pts.contiguous(), off


def refine_links_clusters(path_list, start_steps, end_steps, dof):
    """match clusters_i to clusters_0, in local frame (same signature and files as link.py:85)."""
    dev = _lib.device()
    for link_dir in path_list:
        link_c_files = sorted(glob.glob(link_dir + 'cluster/*.npz'))
        os.makedirs(link_dir + 'cluster_rf', exist_ok=True)
        first = load_pc_npz(link_c_files[start_steps])
        pending = []                                            # (t, n_links, src, src_off) of one shape class

        def flush():
            if not pending:
                return
            k = pending[0][1]
            tgt, toff = _pack(first[:k], dev)
            init = torch.eye(4, dtype=torch.float64, device=dev).repeat(k, 1, 1)
            outs = ops.icp_p2p_batch([(src, soff, tgt, toff, init) for _, _, src, soff in pending], th=1.0,
                                     max_iteration=100000)
            for (t, _, _, soff), (_, moved, _) in zip(pending, outs):
                o, m = soff.cpu().numpy(), moved.cpu().numpy()
                save_pc_npz([m[o[i]:o[i + 1]] for i in range(k)], link_dir + f'cluster_rf/{t:04}.npz')
            pending.clear()

        for t in range(start_steps, end_steps):
            clusters = load_pc_npz(link_c_files[t])
            k = min(dof + 1, len(clusters), len(first))         # zip(range(dof+1), ...) in the reference
            src, soff = _pack(clusters[:k], dev)
            if pending and (pending[0][1] != k or pending[0][2].shape != src.shape or len(pending) == ops.ICP_BATCH_MAX):
                flush()
            pending.append((t, k, src, soff))
        flush()
