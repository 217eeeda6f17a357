"""Experiment: the ICP-style frame with one stream per sequence (no lock-step between sequences) against the lock-step
BatchIcpRegistrar, same frames.  python tests/measure/icp_pipelined.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from autourdf_amd.engine import BatchIcpRegistrar                       # noqa: E402
from autourdf_amd.synthetic import initial_segmentation, make_sequence  # noqa: E402

dev = torch.device("cuda")
S, N, K, F = 5, 4096, 20, 10
seqs = [make_sequence("wx200_5", s, F + 1, N) for s in range(S)]
mats0, clusters0, _ = initial_segmentation(seqs[0][0], K, seed=0)
frames = [[torch.as_tensor(seqs[s][t], dtype=torch.float64, device=dev) for t in range(1, F + 1)] for s in range(S)]


def lockstep():
    reg = BatchIcpRegistrar(mats0, clusters0, S, dev)
    reg.step([frames[s][0] for s in range(S)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(1, F):
        reg.step([frames[s][t] for s in range(S)])
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0), th, [r.M.clone() for r in reg.regs]


def pipelined():
    regs = [BatchIcpRegistrar(mats0, clusters0, 1, dev) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    for s in range(S):
        regs[s].step([frames[s][0]])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(1, F):
        for s in range(S):
            with torch.cuda.stream(streams[s]):
                regs[s].step([frames[s][t]])
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0), th, [r.regs[0].M.clone() for r in regs]


for name, fn in (("lock-step", lockstep), ("one stream per sequence", pipelined), ("lock-step", lockstep), ("one stream per sequence", pipelined)):
    dt, th, M = fn()
    print(f"{name:24s}: {(F - 1) * S / dt:8.1f} frames/s ({dt / (F - 1) * 1e6:7.1f} us per round of {S}; host enqueue {th / (F - 1) * 1e6:7.1f} us)  checksum {float(torch.stack(M).abs().sum()):.9f}")
