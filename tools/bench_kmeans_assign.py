"""K2 E-step micro-benchmark at the BASELINE configs' sizes: VALU vs v_mfma_f64_16x16x4_f64 form of
creg_kmeans_assign_f64 (bit-identical labels).  Prints one JSON line per (n, k, variant) with the
achieved algorithmic GB/s (24 n B read + 4 n B written) and fp64 GFLOP/s (8 n k: 3 FMA + compare).

    python tools/bench_kmeans_assign.py            # on the MI355X
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autourdf_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    shapes = ((4096, 20), (16384, 40), (262144, 128), (1048576, 128))
    if os.environ.get("CREG_KM_SHAPE"):                      # e.g. CREG_KM_SHAPE=1048576,128 for counter passes
        shapes = (tuple(int(v) for v in os.environ["CREG_KM_SHAPE"].split(",")),)
    for n, k in shapes:
        X = torch.as_tensor(rng.normal(size=(n, 3)), device=dev)
        C = X[torch.as_tensor(rng.choice(n, k, replace=False), device=dev)].clone() + 1e-3
        ref = None
        for mfma in (False, True):
            for _ in range(5):
                lab = ops.kmeans_assign(X, C, use_mfma=mfma)
            torch.cuda.synchronize()
            reps = 50
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                lab = ops.kmeans_assign(X, C, use_mfma=mfma)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            ref = lab if ref is None else ref
            print(json.dumps({"kernel": "k_km_assign_mfma" if mfma else "k_km_assign", "n": n, "k": k,
                              "us_per_call_incl_launch": round(us, 2), "algorithmic_GBps": round(28.0 * n / us / 1e3, 1),
                              "fp64_GFLOPs": round(8.0 * n * k / us / 1e3, 1), "labels_equal_valu": bool(torch.equal(lab, ref))}))


if __name__ == "__main__":
    main()
