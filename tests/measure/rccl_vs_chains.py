"""What does a live RCCL process group cost the epoch chains?  (round 5: the default bench line fell 182 -> 112 frames/s the moment
bench.py created a world-1 `nccl` group BEFORE the timed region.)

One process, the configs[1] shape, 5 sequences; the same 6 registered frames are timed
  0. before torch.distributed is touched,
  1. after init_process_group("nccl") WITHOUT a communicator (lazy: no device_id, no collective yet),
  2. after the first collective (communicator, its streams / proxy thread / watchdog exist),
  3. after destroy_process_group().
For every stage: wall per step, device span of a train, host time inside run_batch.

    python tests/measure/rccl_vs_chains.py [eager]      (GPU box; `eager` = init with device_id, as bench.py round 5 first did)
"""
import os
import socket
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
torch.cuda.set_device(0)
S = int(os.environ.get("SEQS", "5"))
seqs = [make_sequence("wx200_5", s, 10, 4096) for s in range(S)]
mats, cl, _ = initial_segmentation(seqs[0][0], 20, seed=0)


def measure(tag):
    reg = BatchRegistrar(mats.astype(np.float32), cl, 4096, S, "q", 512, 300, True, dev, seeds=list(range(S)))
    spans, orig = [], reg._train

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
    for t in range(1, 8):
        frames = [torch.as_tensor(s[t], dtype=torch.float64, device=dev) for s in seqs]
        torch.cuda.synchronize()
        w0 = time.perf_counter()
        reg.step(frames)
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - w0) * 1e3)
    dev_ms = [a.elapsed_time(b) for a, b, _ in spans]
    host_ms = [h for _, _, h in spans]
    w = float(np.median(walls[2:]))
    print(f"{tag:58s} step wall {w:6.2f} ms = {S * 1e3 / w:6.1f} frames/s;  train device span {np.median(dev_ms[4:]):6.2f} ms;  "
          f"host in run_batch {np.median(host_ms[4:]):5.2f} ms", flush=True)
    del reg


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


eager = len(sys.argv) > 1 and sys.argv[1] == "eager"
measure("0 no torch.distributed")
import torch.distributed as dist  # noqa: E402

kw = {"device_id": dev} if eager else {}
dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1, **kw)
measure(f"1 init_process_group(nccl{', device_id' if eager else ''}), no collective yet")
t = torch.zeros(1, device=dev)
dist.all_reduce(t)
torch.cuda.synchronize()
measure("2 after the first all_reduce (communicator built)")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    dist.all_reduce(t)
torch.cuda.synchronize()
measure("2b after an all_reduce issued from a side stream")
dist.destroy_process_group()
torch.cuda.synchronize()
measure("3 after destroy_process_group()")
