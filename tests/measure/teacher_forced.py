"""Teacher-forced single-epoch parity, the statistics behind the tolerances of tests/test_gpu_parity.py::test_train_teacher_forced_*:
    python tests/measure/teacher_forced.py [c1 allegro franka] [--epochs 50,150,299]
For every shape and epoch e: the oracle's state entering e -> ONE epoch of the HIP plan -> against the oracle's epoch e."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np                                                       # noqa: E402
import torch                                                             # noqa: E402
import _teacher as T                                                     # noqa: E402
from autourdf_amd import ops                                             # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden")
SHAPES = {"c1": (20, 4096), "allegro": (30, 4096), "franka": (40, 16384)}
args = [a for a in sys.argv[1:] if not a.startswith("--")]
epochs = (0, 1, 13, 50, 150, 299)
for a in sys.argv[1:]:
    if a.startswith("--epochs="):
        epochs = tuple(int(x) for x in a.split("=", 1)[1].split(","))
dev = torch.device("cuda:0")
c1 = np.load(os.path.join(GOLDEN, "train_reference_c1.npz"))
sd = {k[5:]: torch.from_numpy(c1[k].astype(np.float32)) for k in c1.files if k.startswith("sd16.")}
order = ops.Q_PARAM_ORDER
for shape in (args or list(SHAPES)):
    k, n = SHAPES[shape]
    g = np.load(os.path.join(GOLDEN, f"train_reference_{shape}.npz"))
    t0 = time.perf_counter()
    snaps, hist = T.oracle_snapshots(g, sd, k, epochs)
    print(f"== {shape}: K={k}, N={n}; oracle {max(epochs) + 1} epochs in {time.perf_counter() - t0:.1f} s; reference loss_hist agrees bit for bit: "
          f"{bool(np.array_equal(np.asarray(hist['loss'], np.float64)[:len(g['loss_hist'])], g['loss_hist'][:len(hist['loss'])]))}")
    m, y = torch.from_numpy(g["m"]).to(dev), torch.from_numpy(g["y"]).to(dev)
    pts, off = ops.pack_clusters([torch.from_numpy(c) for c in T.split(g["local"], g["offsets"])], dev)
    plan = ops.TrainPlan("q", k, 512, n, y.shape[0], epochs=300, use_graph=False, device=dev)
    probe_plan = ops.TrainPlan("q", k, 512, n, y.shape[0], epochs=2, use_graph=False, device=dev)
    for e in epochs:
        r = T.compare(order, snaps[e], snaps[e + 1], *T.plan_epoch(plan, probe_plan, dev, order, m, y, pts, off, snaps[e]))
        ex = all(a == b for a, b in r["exact"].values())
        print(f"  e={e:3d}: loss rel {r['loss_rel']:.1e}  pose {r['pose']:.1e}  update |d| {r['upd']:.1e} ({r['upd_worst_tensor']}; unmasked {r['upd_all']:.1e}, masked {100 * r['masked_frac']:.2f} %)  "
              f"exp_avg rel {r['exp_avg_rel']:.1e}  exp_avg_sq rel {r['exp_avg_sq_rel']:.1e}  counters exact {ex}  lr {r['lr'][0]:.6e} / {r['lr'][1]:.6e}  "
              f"lr used exact {r['lr_used_exact']}  sched_best {r['sched_best'][0]:.9e} / {r['sched_best'][1]:.9e}  min_loss {r['min_loss'][0]:.9e} / {r['min_loss'][1]:.9e}"
              + (f"  best_m {r['best_m']:.1e} best_pred {r['best_pred']:.1e}" if 'best_m' in r else ""))
        if not ex:
            print("     counters:", r["exact"])
