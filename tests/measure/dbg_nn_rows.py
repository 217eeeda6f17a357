"""nn_search 0 (sixteen queries per wave) against 2 (four per wave, rounds 2-4) and 1 (exhaustive) on one forward / backward: the pose
gradient depends on the matches' signs and counters only -- it must be IDENTICAL; the loss may differ in its last bits (summation order).
    python tests/measure/dbg_nn_rows.py      (GPU box)
"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from autourdf_amd import ops
from oracle import models
dev = torch.device("cuda:0")
shapes = [("q", 20, 64, 4096, 4096, False), ("dq", 3, 128, 513, 4033, False), ("q", 4, 64, 700, 1, False), ("q", 70, 64, 4096, 4096, False),
          ("q", 150, 64, 9600, 4000, False), ("q", 9, 64, 640, 900, False), ("q", 10, 64, 9000, 6000, False), ("dq", 40, 64, 16384, 16384, False),
          ("q", 3, 64, 500, 5000, False), ("q", 2, 64, 16000, 4097, True), ("q", 5, 64, 3000, 3000, True)]
for rot, k, hidden, n_pred, n_tgt, lattice in shapes:
    g = torch.Generator().manual_seed(n_pred * 7 + n_tgt)
    if lattice:
        y = torch.randint(0, 12, (n_tgt, 3), generator=g).float() * 0.03125
        flat = torch.randint(0, 12, (n_pred, 3), generator=g).float() * 0.03125
    else:
        y = torch.rand(n_tgt, 3, generator=g) * 0.4
        flat = y[torch.randint(0, n_tgt, (n_pred,), generator=g)] + 0.004 * torch.randn(n_pred, 3, generator=g)
    cuts = sorted(torch.randperm(n_pred - 1, generator=g)[: k - 1].add(1).tolist())
    m = torch.eye(4).repeat(k, 1, 1)
    cl = []
    for a, z in zip([0] + cuts, cuts + [n_pred]):
        c = flat[a:z]
        ctr = c.mean(0) if z > a else torch.zeros(3)
        m[len(cl), :3, 3] = ctr
        cl.append(c if lattice else c - ctr)
    if lattice:
        m[:, :3, 3] = 0
    pts, off = ops.pack_clusters(cl, dev)
    torch.manual_seed(5)
    model, order = (models.QRegMLP(True, hidden), ops.Q_PARAM_ORDER) if rot == "q" else (models.DQRegMLP(hidden), ops.DQ_PARAM_ORDER)
    res = {}
    for mode in (0, 2, 1):
        params = [model.state_dict()[key].clone().to(dev) for key in order]
        plan = ops.TrainPlan(rot, k, hidden, n_pred, n_tgt, epochs=2, use_graph=False, device=dev, nn_search=mode)
        m2, pred, loss, grad = plan.probe(m.to(dev), y.to(dev), pts, off, params)
        res[mode] = (float(loss), grad.cpu().numpy(), plan.info)
    l0, g0, i0 = res[0]
    msg = f"{rot} k={k} np={n_pred} nt={n_tgt} lattice={lattice}: "
    for mode in (2, 1):
        l, gg, _ = res[mode]
        msg += f" vs nn_search {mode}: loss rel diff {abs(l - l0) / max(abs(l), 1e-30):.1e} grad identical {np.array_equal(g0, gg)} (max diff {np.abs(g0 - gg).max():.2e});"
    print(msg, flush=True)
