"""BASELINE configs[4] shape (synthetic N=262144 points, K=128 clusters): the assign / re-segmentation step of a frame
-- sklearn-equivalent Lloyd k-means seeded at the pose translations + change of frame (resample_cluster,
mlp_reg.py:172-237) -- on the GPU with the VALU and the matrix-core E-step, next to the C/OpenMP oracle on the host.
Frames are independent at this configuration (200 frames sharded over the ranks).

    python tests/measure/bench_c5_resegment.py > gpurun_out/c5_resegment.log
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from autourdf_amd import ops                                            # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence  # noqa: E402
from oracle import kmeans as okm                                        # noqa: E402  (checker / CPU baseline only)


def main():
    dev = torch.device("cuda")
    N, K, F = 262144, 128, 4
    t0 = time.perf_counter()
    seq = make_sequence("chain32", 0, F + 1, N)
    mats, _, _ = initial_segmentation(seq[0], K, seed=0)
    print(f"# synthetic chain32 sequence: {F + 1} frames of {N} points, {K} poses (host generation {time.perf_counter() - t0:.0f} s)")
    frames = [torch.as_tensor(f, dtype=torch.float64, device=dev) for f in seq[1:]]
    M = torch.as_tensor(mats, dtype=torch.float64, device=dev)
    init = M[:, :3, 3].contiguous()
    ref = {}
    for mfma in (False, True):
        for f in frames[:1]:
            ops.kmeans_lloyd(f, init, use_mfma=mfma)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = []
        for f in frames:
            _, labels, _, n_it = ops.kmeans_lloyd(f, init, use_mfma=mfma)
            local, off = ops.group_to_local(f, labels, M)
            iters.append(n_it)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / len(frames)
        it = float(torch.stack(iters).double().mean())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.kmeans_assign(frames[0], init, use_mfma=mfma)
        e0.record()
        for _ in range(50):
            lab = ops.kmeans_assign(frames[0], init, use_mfma=mfma)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        ref[mfma] = labels.cpu()
        print(f"E-step {'MFMA' if mfma else 'VALU'}: {1.0 / dt:8.1f} frames/s ({dt * 1e3:6.2f} ms per frame, {it:.0f} Lloyd iterations, host-driven convergence test); "
              f"one E-step call {us:6.1f} us incl. launch and the centre pre-pass = {28.0 * N / us / 1e3:6.1f} GB/s algorithmic "
              f"({28.0 * N / us / 1e3 / 8000:.3f} of 8 TB/s), {8.0 * N * K / us / 1e6:5.2f} TFLOP/s fp64")
    print("labels VALU == MFMA:", bool(torch.equal(ref[False], ref[True])))
    t0 = time.perf_counter()
    _, lab_cpu, _, n_cpu = okm.k_means(seq[-1], mats[:, :3, 3])
    cpu_s = time.perf_counter() - t0
    print(f"oracle (C + OpenMP, {os.cpu_count()} host threads available): {cpu_s * 1e3:.0f} ms for the last frame ({n_cpu} iterations); "
          f"labels == GPU: {bool(np.array_equal(lab_cpu, ref[False].numpy()))}")


if __name__ == "__main__":
    main()
