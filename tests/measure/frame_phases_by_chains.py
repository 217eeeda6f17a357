"""Where does a registered frame's time go with one chain (a linear graph, replayed from pre-built packets) and with two chains
(a two-branch graph, every node enqueued by the host)?  Device time of the two trains (events around run_batch) against the
wall-clock of the whole BatchRegistrar.step, 5 sequences at the configs[1] shape.

    python tests/measure/frame_phases_by_chains.py        (GPU box)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import autourdf_amd  # noqa: E402

autourdf_amd.prefer_device_kernargs()
from autourdf_amd.engine import BatchRegistrar  # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence  # noqa: E402

dev = torch.device("cuda:0")
S = 5
seqs = [make_sequence("wx200_5", s, 10, 4096) for s in range(S)]
mats, cl, _ = initial_segmentation(seqs[0][0], 20, seed=0)
def inverse_after_host_wait(self, M, ev):
    """_HostInverse.__call__ with the HOST waiting for the event (no barrier packet parked in the side stream's queue while the
    trains run), then the copy."""
    ev.synchronize()
    with torch.cuda.stream(self.side):
        host = M.to("cpu", non_blocking=True)
        self.side.synchronize()
    inv = np.linalg.inv(host.numpy()).astype(np.float64)
    return torch.from_numpy(inv).to(M.device, non_blocking=True)


from autourdf_amd import engine  # noqa: E402
orig_call = engine._HostInverse.__call__
for gb, stop, hostwait in ((1, 200, False), (1, 200, True), (2, 200, False), (2, 200, True), (-2, 200, False), (-2, 200, True)):
    engine._HostInverse.__call__ = inverse_after_host_wait if hostwait else orig_call
    reg = BatchRegistrar(mats.astype(np.float32), cl, 4096, S, "q", 512, 300, True, dev, seeds=list(range(S)), graph_branches=gb)
    reg.stop = stop
    spans = []
    orig = reg._train

    def timed(problems, lr, same_target=False):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        t0 = time.perf_counter()
        out = orig(problems, lr, same_target)
        t1 = time.perf_counter()
        b.record()
        spans.append((a, b, (t1 - t0) * 1e3))
        return out

    reg._train = timed
    walls = []
    for t in range(1, 9):
        frames = [torch.as_tensor(s[t], dtype=torch.float64, device=dev) for s in seqs]
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        reg.step(frames)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - w0) * 1e3)
    dev_ms = [a.elapsed_time(b) for a, b, _ in spans]
    host_ms = [h for _, _, h in spans]
    print(f"graph_branches {gb:2d} host-side event wait {hostwait!s:5}: step wall {np.median(walls[2:]):6.2f} ms;  train device span {np.median(dev_ms[4:]):6.2f} ms (x2 per step);  "
          f"host in run_batch {np.median(host_ms[4:]):5.2f} ms;  rest of the step {np.median(walls[2:]) - 2 * np.median(dev_ms[4:]):5.2f} ms", flush=True)
    del reg
