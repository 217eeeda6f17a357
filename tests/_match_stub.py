"""Deterministic stand-in for ``train`` (reference mlp_reg.py:17-152) used to PIN THE LOOP BODY OF ``match()``.

``tests/golden/make_golden_match.py`` runs the reference's own ``match()`` (mlp_reg.py:240-386, both branches)
with ``mlp_reg.train`` replaced by this stub, so that what the fixture pins is the composition the reference
performs around ``train`` -- which clusters and poses feed which call, what ``masked_icp`` / ``resample_cluster``
receive, what is written to ``matrix/NNNN.npy`` / ``cluster/NNNN.npz`` / ``loss.txt`` -- not the chaotic 300-epoch
optimisation (pinned per step elsewhere).  The GPU tests replay the same stub through the product's loop
(``mlp_reg.match`` / ``register_sequence`` and ``engine.BatchRegistrar.step`` / ``step_mlp_icp``).

The arithmetic uses only correctly rounded float64 + - * / (no BLAS, no libm, exact ``math.fsum`` sums), so the
stub returns the same bits in the build container (reference run) and on the GPU box (replay).
"""
import math

import numpy as np
import torch


def _fsum_cols(a):
    a = np.asarray(a, np.float64)
    return np.array([math.fsum(a[:, j].tolist()) for j in range(a.shape[1])])


class TrainStub:
    """Callable with ``train``'s signature.  Every call is logged in ``self.calls``:
    (call index, model slot = order in which distinct ``model`` objects were first seen, learning_rate,
    cluster sizes, exact sums of m / y / clusters)."""

    def __init__(self):
        self.calls = []
        self._models = []

    def _slot(self, model):
        for i, m in enumerate(self._models):
            if m is model:
                return i
        self._models.append(model)
        return len(self._models) - 1

    def __call__(self, m, y, model, clusters, stop=200, learning_rate=0.0002, scheduler_patience=5,
                 scheduler_factor=0.7):
        dev = m.device
        call = len(self.calls)
        m64 = m.detach().cpu().numpy().astype(np.float64)
        y64 = y.detach().cpu().numpy().astype(np.float64)
        cl64 = [c.detach().cpu().numpy().astype(np.float64).reshape(-1, 3) for c in clusters]
        K = m64.shape[0]
        self.calls.append(dict(
            call=call, model=self._slot(model), lr=float(learning_rate), sizes=[len(c) for c in cl64],
            sum_m=math.fsum(m64.ravel().tolist()), sum_y=math.fsum(y64.ravel().tolist()),
            sum_c=math.fsum(np.concatenate(cl64).ravel().tolist()) if sum(len(c) for c in cl64) else 0.0))
        gain = 1.0 if learning_rate > 1.5e-4 else 0.5           # "Step" (2e-4) moves further than "Anchor" (1e-4)
        # current world clouds, row by row with separate multiplies and adds
        world = []
        for c, M in zip(cl64, m64):
            w = np.empty_like(c)
            for a in range(3):
                w[:, a] = ((c[:, 0] * M[a, 0] + c[:, 1] * M[a, 1]) + c[:, 2] * M[a, 2]) + M[a, 3]
            world.append(w)
        n_all = sum(len(w) for w in world)
        mean_w = _fsum_cols(np.concatenate(world)) / n_all if n_all else np.zeros(3)
        mean_y = _fsum_cols(y64) / len(y64)
        shift = 0.25 * gain * (mean_y - mean_w)                  # every pose drifts towards the target's centroid
        new_m = m64.copy()
        pred = []
        for k in range(K):
            s = gain * 0.01 * (1 + (k + call) % 3)               # rational rotation about z: no trigonometry
            cs, sn = (1.0 - s * s) / (1.0 + s * s), 2.0 * s / (1.0 + s * s)
            D = np.array([[cs, -sn, 0.0], [sn, cs, 0.0], [0.0, 0.0, 1.0]])
            R = np.empty((3, 3))
            for a in range(3):
                for b in range(3):
                    R[a, b] = (D[a, 0] * m64[k, 0, b] + D[a, 1] * m64[k, 1, b]) + D[a, 2] * m64[k, 2, b]
            t = m64[k, :3, 3] + shift + 0.002 * gain * np.array([1.0, -1.0, 0.5]) * (1 + k % 2)
            new_m[k, :3, :3], new_m[k, :3, 3] = R, t
        new_m32 = new_m.astype(np.float32)                       # train() returns float32 poses
        M32 = new_m32.astype(np.float64)
        for c, M in zip(cl64, M32):                              # ... and the float32 clouds of those poses
            w = np.empty_like(c)
            for a in range(3):
                w[:, a] = ((c[:, 0] * M[a, 0] + c[:, 1] * M[a, 1]) + c[:, 2] * M[a, 2]) + M[a, 3]
            pred.append(w.astype(np.float32))
        min_loss = 0.125 * (call + 1)
        return pred, [None] * K, torch.from_numpy(new_m32).to(dev), min_loss

    def log_arrays(self):
        """The call log as arrays (what the golden stores)."""
        c = self.calls
        return dict(model=np.array([x["model"] for x in c], np.int32), lr=np.array([x["lr"] for x in c]),
                    sizes=np.array([x["sizes"] for x in c], np.int32), sum_m=np.array([x["sum_m"] for x in c]),
                    sum_y=np.array([x["sum_y"] for x in c]), sum_c=np.array([x["sum_c"] for x in c]))
