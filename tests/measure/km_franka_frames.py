"""The re-segmentation of a lock-step round at the franka shape (5 frames of 16384 points, 40 clusters): ONE launch with a workgroup per
frame (k_km_small, what the engine runs) against the many-workgroup Lloyd per frame, one after the other and from 5 host threads on 5
streams.    python tests/measure/km_franka_frames.py     (GPU box)"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from autourdf_amd import ops  # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence  # noqa: E402

dev = torch.device("cuda:0")
S = 5
seqs = [make_sequence("franka", s, 3, 16384) for s in range(S)]
mats, cl, _ = initial_segmentation(seqs[0][0], 40, seed=0)
frames = [torch.as_tensor(s[1], dtype=torch.float64, device=dev) for s in seqs]
inits = [torch.as_tensor(mats[:, :3, 3].copy(), dtype=torch.float64, device=dev) for _ in range(S)]


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


t_batch, ob = timed(lambda: ops.kmeans_lloyd_batch(frames, inits))
t_seq, os_ = timed(lambda: [ops.kmeans_lloyd(f, c) for f, c in zip(frames, inits)])
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]


def threaded():
    outs = [None] * S

    def run(i):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[i]):
            outs[i] = ops.kmeans_lloyd(frames[i], inits[i])
            streams[i].synchronize()
    ths = [threading.Thread(target=run, args=(i,)) for i in range(S)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return outs


t_thr, ot = timed(threaded)
same = all(torch.equal(a[1], b[1]) and torch.equal(a[1], c[1]) for a, b, c in zip(ob, os_, ot))
print(f"5 frames of 16384 points, 40 clusters: one launch, a workgroup per frame {t_batch:.2f} ms | many workgroups per frame, one after the other "
      f"{t_seq:.2f} ms | the same from 5 threads on 5 streams {t_thr:.2f} ms | labels identical: {same}; Lloyd iterations {[int(o[3]) for o in ob]}")
