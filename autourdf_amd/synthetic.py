"""Seeded synthetic articulated-robot point-cloud sequences (host-side data generation, numpy).

No frame data ships with the reference (``data/`` is git-ignored) and its generator
(Sim/sim_data.py) needs PyBullet + OpenGL, so every benchmark / test frame is synthetic
(SURVEY.md §8d).  The noise model and step sizes follow the reference generator:
joint step = step_size * (1 + U[0,1)) degrees per frame (Sim/sim_data.py:417, step_size 4 at :544),
per-frame global translation noise N(0, 0.01^2) m on every frame but the first (:337),
per-point noise N(0, 0.0005^2) m (:343), exactly ``n_points`` points per frame (:347-350).
Seeds: frame t of sequence s uses ``1000 * s + t``.

This is input generation, not the registration path: nothing here runs on the GPU.
"""
import numpy as np

# name -> (list of (parent, length, radius), overall comment).  Serial chains except allegro.
_ROBOTS = {
    # 5-dof arm, extent ~0.4 m  (parameters.json:17-19: num_seg 20, dof 5)
    "wx200_5": [(-1, 0.08, 0.035), (0, 0.10, 0.025), (1, 0.12, 0.022), (2, 0.10, 0.020),
                (3, 0.06, 0.018), (4, 0.05, 0.015)],
    # 7 links, extent ~1 m (parameters.json:33-35)
    "franka": [(-1, 0.18, 0.06), (0, 0.16, 0.055), (1, 0.16, 0.05), (2, 0.14, 0.05),
               (3, 0.14, 0.045), (4, 0.12, 0.04), (5, 0.10, 0.035)],
    # palm + 4 fingers x 3 phalanges, extent ~0.2 m (parameters.json:78-80, key "allegro")
    "allegro": [(-1, 0.09, 0.03)] + [(p, l, 0.011) for f in range(4)
                                     for p, l in ((0, 0.045), (1 + 3 * f, 0.035), (2 + 3 * f, 0.03))],
}
_ROBOTS["allegro_hand"] = _ROBOTS["allegro"]


def _chain(n_links, extent):
    ln = extent / n_links
    return [(i - 1, ln, 0.18 * ln + 0.01) for i in range(n_links)]


def robot_links(robot: str):
    if robot in _ROBOTS:
        return _ROBOTS[robot]
    if robot.startswith("chain"):              # e.g. "chain32" for the N=262144 roofline config
        return _chain(int(robot[5:]), 2.0)
    raise KeyError(f"unknown synthetic robot {robot!r}")


def _rot(axis, ang):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def _capsule_points(rng, n, length, radius):
    """n points on the surface of a capsule along +z from the origin (side + two hemispherical caps)."""
    a_side, a_cap = 2 * np.pi * radius * length, 2 * np.pi * radius ** 2
    which = rng.random(n) * (a_side + 2 * a_cap)
    th = rng.random(n) * 2 * np.pi
    u = rng.random(n)
    pts = np.empty((n, 3))
    side = which < a_side
    pts[side] = np.stack([radius * np.cos(th[side]), radius * np.sin(th[side]), u[side] * length], 1)
    for lo, sign, z0 in ((a_side, -1.0, 0.0), (a_side + a_cap, 1.0, length)):
        m = (which >= lo) & (which < lo + a_cap)
        ct = u[m]                                   # cos(polar) uniform -> uniform on hemisphere
        st = np.sqrt(1 - ct ** 2)
        pts[m] = np.stack([radius * st * np.cos(th[m]), radius * st * np.sin(th[m]),
                           z0 + sign * radius * ct], 1)
    return pts


def make_sequence(robot: str = "wx200_5", seq: int = 0, n_frames: int = 10, n_points: int = 4096,
                  step_size: float = 4.0):
    """Returns a list of ``n_frames`` float64 (n_points, 3) world-frame clouds of one moving robot."""
    links = robot_links(robot)
    L = len(links)
    # geometry and the start configuration belong to the ROBOT (every sequence of the reference
    # starts from the same rest pose, which is why match() may reuse sequence 0's frame-0
    # clustering, mlp_reg.py:242-253); the joint directions belong to the SEQUENCE.
    grng = np.random.default_rng(sum(map(ord, robot)))
    axes = [np.array([1.0, 0, 0]) if i % 2 else np.array([0, 1.0, 0]) for i in range(L)]
    tilt = [_rot(grng.normal(size=3), grng.uniform(-0.3, 0.3)) for _ in range(L)]
    ang = grng.uniform(-0.6, 0.6, size=L)
    ang[0] = 0.0
    srng = np.random.default_rng(977 * seq + 13)
    sign = srng.choice([-1.0, 1.0], size=L)
    area = np.array([2 * np.pi * r * ln + 4 * np.pi * r * r for _, ln, r in links])
    quota = np.floor(area / area.sum() * n_points).astype(int)
    quota[np.argsort(-(area / area.sum() * n_points - quota))[: n_points - quota.sum()]] += 1
    frames = []
    for t in range(n_frames):
        rng = np.random.default_rng(1000 * seq + t)
        if t > 0:
            ang[1:] += sign[1:] * np.deg2rad(step_size * (1 + rng.random(L - 1)))
        Rw, pw = [None] * L, [None] * L
        pts = []
        for i, (parent, ln, rad) in enumerate(links):
            Rl = tilt[i] @ _rot(axes[i], ang[i])
            if parent < 0:
                Rw[i], pw[i] = Rl, np.zeros(3)
            else:
                pl = links[parent][1]
                off = np.array([0, 0, pl])
                if robot.startswith("allegro") and parent == 0:      # spread finger roots on the palm
                    f = (i - 1) // 3
                    off = np.array([0.022 * (f - 1.5), 0.0, pl])
                Rw[i], pw[i] = Rw[parent] @ Rl, pw[parent] + Rw[parent] @ off
            pts.append(_capsule_points(rng, quota[i], ln, rad) @ Rw[i].T + pw[i])
        cloud = np.concatenate(pts)
        if t > 0:
            cloud = cloud + rng.normal(scale=0.01, size=3)
        cloud = cloud + rng.normal(scale=0.0005, size=cloud.shape)
        frames.append(cloud[rng.permutation(n_points)])
    return frames


def kmeans_plusplus(X: np.ndarray, k: int, rng: np.random.Generator) -> np.ndarray:
    """Greedy k-means++ seeding (Arthur & Vassilvitskii) with 2+log(k) local trials, as sklearn's
    ``_kmeans_plusplus`` does for ``init="k-means++"`` (reference cluster_icp.py:67)."""
    n = len(X)
    trials = 2 + int(np.log(k))
    centers = np.empty((k, X.shape[1]))
    centers[0] = X[rng.integers(n)]
    d2 = ((X - centers[0]) ** 2).sum(1)
    for c in range(1, k):
        cand = np.searchsorted(np.cumsum(d2), rng.random(trials) * d2.sum())
        cand = np.clip(cand, 0, n - 1)
        dc = ((X[None, :, :] - X[cand][:, None, :]) ** 2).sum(-1)
        dc = np.minimum(dc, d2[None, :])
        best = dc.sum(1).argmin()
        centers[c], d2 = X[cand[best]], dc[best]
    return centers


def initial_segmentation(frame0: np.ndarray, k: int, seed: int = 0, iters: int = 50):
    """Frame-0 state the way Segments.k_means_cluster builds it (cluster_icp.py:47-107): k-means++
    + Lloyd labels, then per cluster a 4x4 with R = I, t = centroid and the points in that local
    frame.  Host numpy (data preparation for tests / bench; the product's Segments uses the GPU
    Lloyd kernel).  Returns (matrices (K,4,4) f64, list of K local (M_k,3) f64, labels)."""
    X = np.asarray(frame0, np.float64)
    rng = np.random.default_rng(seed)
    C = kmeans_plusplus(X, k, rng)
    lab = None
    for _ in range(iters):
        new = ((X[:, None, :] - C[None, :, :]) ** 2).sum(-1).argmin(1)
        if lab is not None and (new == lab).all():
            break
        lab = new
        for j in range(k):
            if (lab == j).any():
                C[j] = X[lab == j].mean(0)
    mats, clusters = np.tile(np.eye(4), (k, 1, 1)), []
    for j in range(k):
        pts = X[lab == j]
        c = pts.mean(0) if len(pts) else C[j]
        mats[j, :3, 3] = c
        clusters.append(pts - c)
    return mats, clusters, lab.astype(np.int32)
