"""Phase-by-phase wall time of the ICP-style frame (K3 -> K4 -> K5 -> K2 -> change of frame), each phase bracketed by
a device synchronisation, at the configs[1] shape (5 sequences in lock-step) and the configs[4] shape (N=262144, K=128).

    python tests/measure/profile_icp_frame.py [c1|c5|both] > gpurun_out/icp_frame_phases.log
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from autourdf_amd import ops                                            # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence  # noqa: E402


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


def c5(n_frames=6):
    dev = torch.device("cuda")
    N, K = 262144, 128
    seq = make_sequence("chain32", 0, n_frames + 1, N)
    mats0, clusters0, _ = initial_segmentation(seq[0], K, seed=0, iters=8)
    M = torch.as_tensor(mats0, dtype=torch.float64, device=dev).contiguous()
    local, off = ops.pack_clusters(clusters0, dev, torch.float64)
    print(f"# configs[4] shape: N={N}, K={K}; per-frame phase times in ms (device synchronised around every phase)")
    for t, f in enumerate(seq[1:]):
        f64 = torch.as_tensor(f, dtype=torch.float64, device=dev)
        world32, t_k3 = timed(lambda: ops.cluster_transform(local.to(torch.float32), off, M.to(torch.float32)))
        (M_new, _, n_it), t_k4 = timed(lambda: ops.masked_icp(local, world32, off, f64, M))
        _, t_k5 = timed(lambda: ops.se3_to_dq(M_new.to(torch.float32)))
        (_, labels, _, n_km), t_k2 = timed(lambda: ops.kmeans_lloyd(f64, M_new[:, :3, 3].contiguous()))
        (local, off), t_grp = timed(lambda: ops.group_to_local(f64, labels, M_new))
        M = M_new
        it = n_it.cpu().numpy()
        sizes = np.diff(off.cpu().numpy())
        print(f"frame {t}: clusters of {sizes.min()}..{int(np.median(sizes))}..{sizes.max()} points (min / median / max); K3 {t_k3:6.2f}  K4 {t_k4:7.2f} (iterations mean {it.mean():5.1f} max {it.max():3d}, clusters over 40: {(it > 40).sum()})  "
              f"K5 {t_k5:5.2f}  K2 {t_k2:6.2f} ({int(n_km)} Lloyd iterations)  change of frame {t_grp:5.2f}  "
              f"sum {t_k3 + t_k4 + t_k5 + t_k2 + t_grp:7.2f}")


def c1(n_frames=8, S=5):
    from autourdf_amd.engine import BatchIcpRegistrar
    dev = torch.device("cuda")
    N, K = 4096, 20
    seqs = [make_sequence("wx200_5", s, n_frames + 1, N) for s in range(S)]
    mats0, clusters0, _ = initial_segmentation(seqs[0][0], K, seed=0)
    reg = BatchIcpRegistrar(mats0, clusters0, S, dev)
    print(f"# configs[1] shape: N={N}, K={K}, {S} sequences in lock-step; per-round phase times in us")
    for t in range(1, n_frames + 1):
        frames = [torch.as_tensor(seqs[s][t], dtype=torch.float64, device=dev) for s in range(S)]
        icp, t_k4 = timed(lambda: ops.masked_icp_batch([(r.local, None, r.off, f, r.M) for r, f in zip(reg.regs, frames)]))
        M_all = torch.stack([m for m, _, _ in icp])
        _, t_k5 = timed(lambda: ops.se3_to_dq(M_all.to(torch.float32).reshape(-1, 4, 4)))
        inits = [M_all[i, :, :3, 3].contiguous() for i in range(S)]
        km, t_k2 = timed(lambda: ops.kmeans_lloyd_batch(frames, inits))
        groups, t_grp = timed(lambda: ops.group_to_local_batch(frames, [kr[1] for kr in km], [o[0] for o in icp]))
        for r, o, (local, off) in zip(reg.regs, icp, groups):
            r.local, r.off, r.M = local, off, o[0]
        it = torch.stack([o[2] for o in icp]).cpu().numpy()
        kmi = [int(kr[3]) for kr in km]
        print(f"round {t}: K4 {t_k4 * 1e3:7.1f} (iterations mean {it.mean():5.1f} max {it.max():3d})  K5 {t_k5 * 1e3:6.1f}  "
              f"K2 {t_k2 * 1e3:6.1f} (Lloyd iterations {kmi})  change of frame {t_grp * 1e3:6.1f}")
    # the unsynchronised round, as bench.py times it
    frames_all = [[torch.as_tensor(seqs[s][t], dtype=torch.float64, device=dev) for s in range(S)] for t in range(1, n_frames + 1)]
    reg = BatchIcpRegistrar(mats0, clusters0, S, dev)
    reg.step(frames_all[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for fr in frames_all[1:]:
        reg.step(fr)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"unsynchronised: {t_all / (n_frames - 1) * 1e6:7.1f} us per round of {S} frames (host enqueue alone {t_host / (n_frames - 1) * 1e6:7.1f} us)")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    if which in ("c1", "both"):
        c1()
    if which in ("c5", "both"):
        c5()
