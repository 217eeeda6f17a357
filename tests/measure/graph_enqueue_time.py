"""Host time to ENQUEUE a 300-epoch train (6 graph replays) against the device time it takes, per launch mode.

    python tests/measure/graph_enqueue_time.py            (GPU box; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0/1 from the environment)

graph_branches 1 / 2 / 3: one captured graph with that many parallel chains; -2 / -3: chain-stream mode (every chain its own linear
graph on its own stream).  If the host needs as long to feed the queues as the device needs to drain them, the train is host-bound and
its rate depends on how the host thread is scheduled -- the suspected cause of the run-to-run modes of three chains.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from autourdf_amd import ops  # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence  # noqa: E402
from oracle import models  # noqa: E402  (parameter shapes / init only)

dev = torch.device("cuda:0")
seq = make_sequence("wx200_5", 0, 3, 4096)
mats, cl, _ = initial_segmentation(seq[0], 20, seed=0)
m = torch.tensor(mats, dtype=torch.float32, device=dev)
ys = [torch.tensor(seq[1] + 0.001 * b, dtype=torch.float32, device=dev) for b in range(5)]
pts, off = ops.pack_clusters([torch.tensor(c, dtype=torch.float32) for c in cl], dev)
torch.manual_seed(0)
model = models.QRegMLP(True, 512)
mk = lambda: [model.state_dict()[k].clone().to(dev) for k in ops.Q_PARAM_ORDER]
print("DEBUG_CLR_GRAPH_PACKET_CAPTURE =", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "(unset)"))
for gb in (1, 2, 3, -2, -3):
    plan = ops.TrainPlan("q", 20, 512, pts.shape[0], 4096, epochs=300, use_graph=True, device=dev, batch=5, graph_branches=gb)
    probs = [(m, ys[b], pts, off, mk()) for b in range(5)]
    plan.run_batch(probs, stop=10 ** 6)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(8):
        t0 = time.perf_counter()
        plan.run_batch(probs, stop=10 ** 6)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3)
        tot.append((t2 - t0) * 1e3)
    print(f"graph_branches {gb:2d}: host enqueue {np.median(enq):6.2f} ms (min {min(enq):.2f}, max {max(enq):.2f})   "
          f"train done after {np.median(tot):6.2f} ms (min {min(tot):.2f}, max {max(tot):.2f})", flush=True)
    del plan
