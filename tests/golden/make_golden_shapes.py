"""Mint tests/golden/train_reference_{allegro,franka}.npz (run in the BUILD CONTAINER only): the REFERENCE's own `train()`
(/root/reference/PointCloud/mlp_reg.py:17-152) with the default model as `match()` builds it (`QRegMLP(True, hidden_dim=512)`,
mlp_reg.py:281-282) at the two other registration shapes of BASELINE.json -- configs[3] allegro (N = 4096, K = 30) and configs[2]
franka (N = 16384, K = 40) -- on the bench's synthetic frames (autourdf_amd.synthetic), imported through tests/golden/ref_shims.py.

    python tests/golden/make_golden_shapes.py [allegro|franka ...]

VERDICT r4 ("what's weak" 1): the whole-frame parity claim -- min_loss / best pose of a 300-epoch train inside twice the spread of
correct float32 implementations -- rested on ONE problem (train_reference_c1.npz).  These two fixtures put the other shapes under
the same test.  What is pinned per shape:
  * inputs (poses, target frame, local clusters + offsets, float32); the state_dict is NOT stored again: QRegMLP's parameters do not
    depend on K, so the float16-rounded state of train_reference_c1.npz (`sd16.*`) is what the run starts from;
  * one 300-epoch run of the reference function: the loss of EVERY epoch, the pose matrix `calculate_pc` received at the epochs in
    `pose_epochs` (0..16, then 20 / 30 / 50 / 100 / 150 / 200 / 250 / 299), best_m, min_loss;
  * one 6-epoch run: losses, best_m, best cloud (the 1e-5 regime).
The third-party arithmetic inside (pytorch3d's knn / quaternion maps) is the oracle's restatement (ref_shims): what this pins is the
reference's composition at these shapes.
"""
import builtins
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
import model_utils as ref_models  # noqa: E402  (reference)
import mlp_reg as ref_reg  # noqa: E402  (reference)

sys.path.insert(0, "/root/repo")
from autourdf_amd.synthetic import initial_segmentation, make_sequence  # noqa: E402

SHAPES = {"allegro": ("allegro", 4096, 30), "franka": ("franka", 16384, 40)}
POSE_EPOCHS = tuple(range(17)) + (20, 30, 50, 100, 150, 200, 250, 299)
HIDDEN = 512


def mint(shape, out_dir=HERE):
    robot, n, k = SHAPES[shape]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    c1 = np.load(os.path.join(HERE, "train_reference_c1.npz"))
    sd16 = {key[5:]: torch.from_numpy(c1[key]) for key in c1.files if key.startswith("sd16.")}
    seq = make_sequence(robot, seq=0, n_frames=2, n_points=n)
    mats, clusters, _ = initial_segmentation(seq[0], k, seed=0)
    mats = mats.astype(np.float32)
    clusters = [c.astype(np.float32) for c in clusters]
    y = torch.from_numpy(seq[1].astype(np.float32))
    ref_reg.ROT = "q"
    out = {"m": mats, "y": y.numpy(), "local": np.concatenate(clusters), "offsets": np.cumsum([0] + [len(c) for c in clusters]),
           "hidden": np.int64(HIDDEN), "pose_epochs": np.array(POSE_EPOCHS, np.int64)}
    for n_ep in (6, 300):
        model = ref_models.QRegMLP(True, hidden_dim=HIDDEN)
        model.load_state_dict({kk: v.to(torch.float32) for kk, v in sd16.items()})
        losses, poses = [], []
        orig_cd, orig_pc = ref_reg.chamfer_distance, ref_reg.calculate_pc

        def spy_cd(*a, **kw):
            r = orig_cd(*a, **kw)
            losses.append(float(r[0].item()))
            return r

        def spy_pc(local_clusters, matrices):
            poses.append(matrices.detach().clone().numpy())
            return orig_pc(local_clusters, matrices)

        ref_reg.chamfer_distance, ref_reg.calculate_pc = spy_cd, spy_pc
        ref_reg.range = lambda nn, _n=n_ep: builtins.range(_n if nn == 300 else nn)     # only the epoch loop (mlp_reg.py:60)
        try:
            pred_np, _, best_m, min_loss = ref_reg.train(torch.from_numpy(mats), y, model, [torch.from_numpy(c) for c in clusters])
        finally:
            ref_reg.chamfer_distance, ref_reg.calculate_pc = orig_cd, orig_pc
            del ref_reg.range
        tag = f"e{n_ep}"
        out.update({f"{tag}_best_m": best_m.detach().numpy(), f"{tag}_min_loss": np.float64(min_loss)})
        if n_ep == 300:
            out["loss_hist"] = np.array(losses, np.float64)
            out["pose_hist_sel"] = np.stack([poses[e] for e in POSE_EPOCHS]).astype(np.float32)
        else:
            out[f"{tag}_loss_hist"] = np.array(losses, np.float64)
            out[f"{tag}_best_pred"] = np.concatenate(pred_np).astype(np.float32)
        print(shape, tag, "min_loss", min_loss, "epochs", len(losses), flush=True)
    path = os.path.join(out_dir, f"train_reference_{shape}.npz")
    np.savez_compressed(path, **out)
    print(f"{os.path.basename(path)} {os.path.getsize(path) / 1024:8.1f} KB")


if __name__ == "__main__":
    for s in (sys.argv[1:] or list(SHAPES)):
        mint(s)
