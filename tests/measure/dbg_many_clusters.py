"""Where does the plan's first-epoch weight gradient differ from torch autograd on the oracle at many clusters?  (round 5)
    python tests/measure/dbg_many_clusters.py        (GPU box)
"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from autourdf_amd import ops
from autourdf_amd.synthetic import initial_segmentation, make_sequence
from oracle import registration
from oracle.chamfer import chamfer_distance
import test_gpu_parity as T
dev = torch.device("cuda:0")
for rot, k in (("6d", 144), ("6d", 160), ("q", 160), ("rpy", 160), ("6d", 100)):
    seq = make_sequence("franka", 23, 2, 16384)
    mats, cl, _ = initial_segmentation(seq[0], k, seed=4)
    m = torch.tensor(mats, dtype=torch.float32); y = torch.tensor(seq[1], dtype=torch.float32)
    clusters = [torch.tensor(c, dtype=torch.float32) for c in cl]
    torch.manual_seed(11)
    model = T._oracle_model(rot, 256); order = T._order(rot); sd = model.state_dict()
    params = [sd[n].clone().to(dev) for n in order]
    pts, off = ops.pack_clusters(clusters, dev)
    plan = ops.TrainPlan(rot, k, 256, pts.shape[0], y.shape[0], epochs=1, use_graph=False, device=dev)
    plan.run(m.to(dev), y.to(dev), pts, off, params, lr=1e-3)
    torch.cuda.synchronize()
    mom = T._plan_first_moments(plan, order, {n: tuple(sd[n].shape) for n in order})
    keep = {}
    def hook(mod, i, o):
        o.retain_grad(); keep["pre"] = o; keep["enc"] = i[0].detach()
    h = model.encoder[0].register_forward_hook(hook)
    m2 = registration.pose_forward(m, model, rot)
    pred = torch.cat(registration.calculate_pc(clusters, m2))
    loss, _ = chamfer_distance(pred.unsqueeze(0), y.unsqueeze(0), norm=1)
    loss.backward(); h.remove()
    named = dict(model.named_parameters())
    print(f"== {rot} k={k}")
    for name in order:
        g_ref = named[name].grad.numpy(); g_plan = mom[name] / np.float32(0.1)
        e = np.abs(g_plan - g_ref); print(f"   {name:22s} max err {e.max():.3e}  max |g| {np.abs(g_ref).max():.3e}  ratio {e.max()/np.abs(g_ref).max():.2e}")
    # the encoder: which pose rows' contributions explain the residual?
    gpre = keep["pre"].grad.numpy().astype(np.float64)        # [K, H]
    enc = keep["enc"].numpy().astype(np.float64)              # [K, IN]
    resid = (mom["encoder.0.weight"] / np.float32(0.1)).astype(np.float64) - named["encoder.0.weight"].grad.numpy().astype(np.float64)   # [H, IN]
    f64 = gpre.T @ enc
    print(f"   encoder: |plan - f64| max {np.abs(mom['encoder.0.weight'] / np.float32(0.1) - f64).max():.3e}   |torch - f64| max {np.abs(named['encoder.0.weight'].grad.numpy() - f64).max():.3e}")
    # project the residual on every row's outer product
    coef = np.array([(resid * np.outer(gpre[r], enc[r])).sum() / max((np.outer(gpre[r], enc[r]) ** 2).sum(), 1e-300) for r in range(k)])
    top = np.argsort(-np.abs(coef))[:6]
    print("   rows whose contribution best explains the residual (row: coefficient): " + "  ".join(f"{r}: {coef[r]:+.2e}" for r in top))
    col = np.abs(resid).max(0); print("   residual max by feature column (first 8, last 8): ", np.round(col[:8] / np.abs(f64).max(), 7), np.round(col[-8:] / np.abs(f64).max(), 7))
