"""The drop-in command line end to end at the configs[1] shape: `python -m autourdf_amd.mlp_reg --robot wx200_5` on a synthetic data
directory (5 sequences x 10 frames of N = 4096 points as binary PLY, 20 clusters) -- wall clock per registered frame, PLY parsing,
frame-0 k-means and the matrix / cluster files included, beside bench.py's 5.5 ms (which times the registration alone).

    python tests/measure/cli_end_to_end.py [frames] [sequences]      (GPU box; writes under a temporary directory)
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from autourdf_amd.synthetic import make_sequence  # noqa: E402


def write_ply(path, pts):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\n"
                b"property double z\nend_header\n" % len(pts))
        f.write(np.ascontiguousarray(pts, "<f8").tobytes())


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seqs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    tmp = tempfile.mkdtemp(prefix="creg_cli_")
    for v in range(seqs):
        for t, fr in enumerate(make_sequence("wx200_5", v, frames, 4096)):
            write_ply(f"{tmp}/data/raw/wx200_5/4_deg_20_cams/V{v:04}/{t:04}/robot.ply", fr)
    json.dump({"wx200_5": {"num_seg": 20, "dof": 5}}, open(f"{tmp}/parameters.json", "w"))
    os.chdir(tmp)
    from autourdf_amd import mlp_reg
    import torch
    for rep in range(2):                                   # the second run reuses the plans and the warmed-up runtime
        for d in ("data/part",):
            if os.path.isdir(d):
                import shutil
                shutil.rmtree(d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mlp_reg.main(["--robot", "wx200_5", "--num_video", str(seqs)])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = seqs * (frames - 1)
        print(f"run {rep}: {dt:.3f} s for {n} registered frames of {seqs} sequences = {dt / n * 1e3:.2f} ms per frame "
              f"({n / dt:.1f} frames/s end to end)", flush=True)


if __name__ == "__main__":
    main()
