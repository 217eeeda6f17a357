"""Blocks tested / visited per wave by the sixteen-queries-per-wave search (build: CREG_EXTRA_FLAGS=-DCREG_NN_STATS python -m
autourdf_amd.build --variant nnstats; run: CREG_LIB_VARIANT=nnstats CREG_NN_ROWS=1 python tests/measure/nn_rows_stats.py)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from autourdf_amd import _lib, ops
from autourdf_amd.synthetic import initial_segmentation, make_sequence
from oracle import models
dev = torch.device("cuda:0")
L = _lib.load()
L.creg_debug_nn_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
for robot, n, k in (("wx200_5", 4096, 20), ("allegro", 4096, 30), ("franka", 16384, 40)):
    seq = make_sequence(robot, 0, 2, n)
    mats, cl, _ = initial_segmentation(seq[0], k, seed=0)
    m = torch.tensor(mats, dtype=torch.float32, device=dev); y = torch.tensor(seq[1], dtype=torch.float32, device=dev)
    pts, off = ops.pack_clusters([torch.tensor(c, dtype=torch.float32) for c in cl], dev)
    torch.manual_seed(0)
    model = models.QRegMLP(True, 512)
    params = [model.state_dict()[key].clone().to(dev) for key in ops.Q_PARAM_ORDER]
    plan = ops.TrainPlan("q", k, 512, pts.shape[0], n, epochs=2, use_graph=False, device=dev)
    out = (ctypes.c_ulonglong * 8)()
    plan.probe(m, y, pts, off, params); torch.cuda.synchronize()
    L.creg_debug_nn_stats(out, 1)
    plan.probe(m, y, pts, off, params); torch.cuda.synchronize()
    L.creg_debug_nn_stats(out, 1)
    for d, name in ((0, "predicted -> target"), (1, "target -> predicted")):
        w, c, v = out[4 * d], out[4 * d + 1], out[4 * d + 2]
        print(f"{robot} N={n} K={k} {name}: {w} waves with queries, blocks tested per wave {c / max(w, 1):.2f}, visited per wave {v / max(w, 1):.2f}", flush=True)
