"""Debug build only: where the cycles of one k_masked_icp workgroup go (see CREG_STAMPS in csrc/icp.hip).

    CREG_EXTRA_FLAGS=-DCREG_STAMPS python -m autourdf_amd.build --force
    python tests/measure/icp_stamps.py > gpurun_out/icp_stamps.log
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from autourdf_amd import _lib, ops                                      # noqa: E402
from autourdf_amd.engine import BatchIcpRegistrar                       # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence  # noqa: E402

NAMES = ["mask+setup", "NN scan", "combine + mean sums", "covariance sums", "Horn on lane 0", "move + barrier", "iterations", "targets scanned"]


def main():
    L = _lib.load()
    fn = L.creg_debug_icp_stamps
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    dev = torch.device("cuda")
    S, N, K = 5, 4096, 20
    seqs = [make_sequence("wx200_5", s, 6, N) for s in range(S)]
    mats0, clusters0, _ = initial_segmentation(seqs[0][0], K, seed=0)
    reg = BatchIcpRegistrar(mats0, clusters0, S, dev)
    out = (ctypes.c_ulonglong * (512 * 16))()
    for t in range(1, 6):
        frames = [torch.as_tensor(seqs[s][t], dtype=torch.float64, device=dev) for s in range(S)]
        torch.cuda.synchronize()
        fn(None, 1)
        res = reg.step(frames)
        torch.cuda.synchronize()
        fn(out, 0)
        allv = np.array(list(out), dtype=np.float64).reshape(512, 16)
        b = int(allv[:, 6].argmax())
        v = allv[b]
        it = max(v[6], 1)
        print(f"round {t}: slowest workgroup {b} (cluster {b % K} of sequence {b // K}): {int(v[6])} iterations; total cycles {v[:6].sum():.0f} = {v[:6].sum() / 2400:.0f} us at 2.4 GHz; per iteration:")
        for i in range(1, 6):
            print(f"    {NAMES[i]:28s} {v[i] / it:9.0f}  ({v[i] / it / 2400:6.2f} us at 2.4 GHz)")
        print(f"    NN scan split (wave 0): bounds + range {v[8] / (it + 1):.0f}, scan {v[9] / (it + 1):.0f}, wait for the other waves {v[11] / (it + 1):.0f} cycles per search; rescans for ties: {int(v[10])}")
        print(f"    {NAMES[0]:28s} {v[0]:9.0f} once ({v[0] / 2400:6.2f} us);  targets scanned per wave-unit and correspondence search: {v[7] / (it + 1):.0f} (all units of the workgroup summed)")


if __name__ == "__main__":
    main()
