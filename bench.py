"""bench.py -- registered frames/sec of the cluster-registration hot path on MI355X.

One "step" = one registered frame exactly as the reference's default path does it
(mlp_reg.py:334-378): train "Step" (300 Adam epochs) + train "Anchor" (300 epochs, lr 1e-4) +
resample_cluster (Lloyd k-means + change of frame), at N=4096 points, K=20 clusters, QRegMLP
hidden 512 (BASELINE.json configs[1]: wx200_5-shaped, 5 sequences x 10 frames).

Two ways of spreading frames over GPUs (one process per GPU, no data-path collective, ONE final
all_gather of the (frames,K,4,4) poses over RCCL):

  --mode sequences (default, the driver's line)  frames of a sequence are sequentially dependent
        (mlp_reg.py:293-378), sequences are not: every rank registers its own S sequences in
        lock-step.  WEAK scaling: --steps frames per rank.
  --mode replay   independent-frame mode (what "50 frames sharded
        across 8 GPUs" must mean): a work item = (poses_t, clusters_t, clusters_0, frame_t+1, both
        models' weights) captured from a sequential pass that every rank repeats untimed; the --steps
        items of the job are dealt round-robin to the ranks and registered independently.  STRONG
        scaling: --steps frames in total.  --workload c5 (N=262144, K=128: SURVEY 8(d) "assign / fit
        kernels only") always runs this way, with the ICP-style frame (K4 masked ICP + K5 + K2).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode replay] [--workload allegro|franka|c5]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import time

# kernel arguments in device memory: this image's HIP runtime does so by default; with HIP_FORCE_DEV_KERNARG=0 every launch reads its
# arguments from host memory and the headline measured 123 instead of 149 frames/s.  Only a default -- an explicit setting is respected.
# (An entry point sets it, before the HIP runtime starts; importing autourdf_amd does not touch the environment.)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# dmabuf IPC only on this host driver (RCCL / cross-process device memory needs it): set here as well as by the self-launcher, so that a
# launcher the DRIVER supplies (torch.distributed.run started by someone else) gets it too
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HIDDEN, EPOCHS, FRAMES_PER_SEQ = 512, 300, 10
MODEL_NAMES = {"q": "QRegMLP hidden 512", "dq": "--r dq: DQRegMLP hidden 512", "6d": "--r 6d: RRegMLP hidden 512",
               "rpy": "--r rpy: RegMLP(6, 3), hidden 3 as mlp_reg.py:285 builds it"}
# QRegMLP(multi_decoder=True, hidden 512) parameters: 56->512, 512->256->3, 512->512->4 with biases
N_PARAMS = (56 * HIDDEN + HIDDEN) + (HIDDEN * (HIDDEN // 2) + HIDDEN // 2) + (3 * (HIDDEN // 2) + 3) + (HIDDEN * HIDDEN + HIDDEN) + (4 * HIDDEN + 4)
# BASELINE.json configs: [1] is the headline (default); the others are selectable for extra evidence lines
WORKLOADS = {"wx200_5": ("wx200_5", 4096, 20, "BASELINE configs[1]"),
             "franka": ("franka", 16384, 40, "BASELINE configs[2] shape"),
             "allegro": ("allegro_hand", 4096, 30, "BASELINE configs[3] shape"),
             "c5": ("chain32", 262144, 128, "BASELINE configs[4] shape (ICP-style frame: assign + fit kernels)"),
             # round 6: frames of the REAL robots (the reference's URDFs + meshes through the sim_data path, minted in the build container by
             # tests/golden/make_golden_real_frames.py: 2 sequences x 10 frames x 4096 points each) instead of the capsule chains of synthetic.py
             "wx200_5_real": ("wx200_5_real", 4096, 20, "BASELINE configs[1] on real wx200 geometry: tests/golden/frames_wx200_5_real.npz"),
             "franka_real": ("franka_real", 4096, 20, "real franka_panda geometry at N=4096 / num_seg 20: tests/golden/frames_franka_real.npz")}
HBM_PEAK_GBPS = 8000.0
STUB = os.environ.get("CREG_BENCH_STUB") == "1"      # CPU plumbing test: gloo + a stand-in registrar (tests/test_bench_cpu.py)


# ------------------------------------------------------------------------------------------ CPU baseline
_ALL_THREADS_CHILD = r"""
import os, sys, time
n = int(sys.argv[2]); os.environ["OMP_NUM_THREADS"] = str(n)
sys.path.insert(0, sys.argv[3])
import numpy as np, torch
torch.set_num_threads(n)
from oracle import _clib, models, registration
_clib.lib().oracle_set_threads(n)
d = np.load(sys.argv[1])
torch.manual_seed(0)
m, y = torch.tensor(d["m"]), torch.tensor(d["y"])
off = d["off"]; cl = [torch.tensor(d["pts"][off[i]:off[i + 1]]) for i in range(len(off) - 1)]
print("READY", flush=True)
for ep in (1, 2):
    t = time.perf_counter()
    registration.train(m, y, models.QRegMLP(True, int(sys.argv[4])), cl, rot="q", epochs=ep)
    print("DONE", ep, time.perf_counter() - t, flush=True)
"""


def _all_threads_probe(m, y, cl0, ncpu, cap_s=3.0):
    """The port on ALL hardware threads of the box ("all host cores and 1") -- the pathological team size for latency-sized
    work (a 256-thread fork-join per tensor op: one cold epoch took 17 s of the driver's 42 s run in round 5).  Run in a CHILD process and
    abandoned `cap_s` seconds after the child has finished importing: a reported data point must not cost more than it is worth."""
    import subprocess
    import sys
    import tempfile
    import threading
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "p.npz")
        off = np.cumsum([0] + [len(c) for c in cl0])
        np.savez(path, m=m.numpy(), y=y.numpy(), pts=np.concatenate([c.numpy() for c in cl0]).astype(np.float32), off=off)
        p = subprocess.Popen([sys.executable, "-c", _ALL_THREADS_CHILD, path, str(ncpu), os.path.dirname(os.path.abspath(__file__)), str(HIDDEN)],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        lines, ready = [], threading.Event()

        def reader():
            for ln in p.stdout:
                lines.append(ln.split())
                ready.set()

        th = threading.Thread(target=reader, daemon=True)
        th.start()
        t_import = time.perf_counter()
        while not any(l and l[0] == "READY" for l in lines) and p.poll() is None and time.perf_counter() - t_import < 90.0:
            time.sleep(0.05)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < cap_s and p.poll() is None:
            time.sleep(0.05)
        if p.poll() is None:
            p.kill()
        p.wait()
        th.join(timeout=2.0)
    done = {int(l[1]): float(l[2]) for l in lines if l and l[0] == "DONE"}
    if 2 in done:
        return done[2] / 2, "two epochs after one untimed epoch, in a child process"
    if 1 in done:
        return done[1], f"ONE cold epoch of {done[1]:.2f} s in a child process (the second did not finish inside the {cap_s:.0f} s cap)"
    return None, f"abandoned: not even one epoch inside the {cap_s:.0f} s cap (child process killed)"


def cpu_baseline(seq0, mats0, clusters0, n_points, k_clusters, budget_frames=2, budget_s=25.0):
    """The oracle (CPU port of the reference path, `kind: "port"`) on the host cores.  The team sizes 1 / 8 / 16 / 32 are sampled (one untimed
    epoch, then three); the two fastest each register FULL frames (2 x 300 Adam epochs + the resample, SURVEY 8(d): two frames for the
    fastest by sample, one for the runner-up, inside a 25 s budget) and `value` is the faster sustained rate -- VERDICT r5 weak 6: the divisor
    is the fastest configuration observed, never a hard-coded team.  64 threads are sampled afterwards, all threads in a child process under
    a 3 s cap."""
    from oracle import _clib, models, registration
    torch.manual_seed(0)
    ncpu = os.cpu_count() or 1

    def team(nt):
        os.environ["OMP_NUM_THREADS"] = str(nt)
        torch.set_num_threads(nt)
        _clib.lib().oracle_set_threads(nt)

    m = torch.tensor(mats0, dtype=torch.float32)
    cl = [torch.tensor(c, dtype=torch.float32) for c in clusters0]
    cl0 = [c.clone() for c in cl]
    y1 = torch.tensor(seq0[1], dtype=torch.float32)
    team(min(16, ncpu))
    registration.train(m, y1, models.QRegMLP(True, HIDDEN), cl, rot="q", epochs=2)   # warm caches / build the C lib
    team_points = {}

    def time_team(nt):
        team(nt)
        registration.train(m, y1, models.QRegMLP(True, HIDDEN), cl0, rot="q", epochs=1)
        tt = time.perf_counter()
        n_ep = 3 if nt > 1 else 6
        registration.train(m, y1, models.QRegMLP(True, HIDDEN), cl0, rot="q", epochs=n_ep)
        team_points[nt] = (time.perf_counter() - tt) / n_ep

    for nt in (1, 8, 16, 32):                              # (ascending; 64 and all threads come after `value` is measured)
        if nt <= ncpu:
            time_team(nt)
    # `value` = SUSTAINED full frames, at the fastest team.  A 3-epoch sample does not predict a 600-epoch frame (32 threads sampled 4.8 ms per
    # epoch and then ran two frames at 10.9 on the GPU box), so the two fastest teams by sample each register full frames and the faster one is
    # `value` (the divisor must be the fastest configuration observed, not a hard-coded 16).
    def full_frames(nt, n_frames):
        team(nt)
        model, model_rf = models.QRegMLP(True, HIDDEN), models.QRegMLP(True, HIDDEN)
        mm, cc = torch.tensor(mats0, dtype=torch.float32), [c.clone() for c in cl0]
        t0 = time.perf_counter()
        ep_done, fr_done, t_km = 0, 0, 0.0
        for f in range(n_frames):
            y = torch.tensor(seq0[f + 1], dtype=torch.float32)
            if f > 0 and budget_s / 2 - (time.perf_counter() - t0) < (time.perf_counter() - t0) / max(fr_done, 1):
                break
            _, m1, _, h1 = registration.train(mm, y, model, cc, rot="q", epochs=EPOCHS)
            _, m2, _, h2 = registration.train(m1.detach(), y, model_rf, cl0, rot="q", epochs=EPOCHS, learning_rate=1e-4)
            ep_done += len(h1["loss"]) + len(h2["loss"])
            tk = time.perf_counter()
            new, _ = registration.resample_cluster(seq0[f + 1], k_clusters, m2.detach().numpy())
            t_km += time.perf_counter() - tk
            mm, cc = m2.detach(), [torch.tensor(c, dtype=torch.float32) for c in new]
            fr_done += 1
        el = time.perf_counter() - t0
        return {"threads": nt, "elapsed": el, "frames": fr_done, "epochs": ep_done, "t_km": t_km, "frame_s": el / max(fr_done, 1)}

    ranked = sorted((nt for nt in team_points if nt > 1), key=team_points.get)[:2] or [1]
    runs = [full_frames(nt, budget_frames if i == 0 else 1) for i, nt in enumerate(ranked)]
    best = min(runs, key=lambda r: r["frame_s"])
    threads, elapsed, frames_done, epochs_done, t_km = best["threads"], best["elapsed"], best["frames"], best["epochs"], best["t_km"]
    team(threads)
    per_epoch = (elapsed - t_km) / max(epochs_done, 1)
    frame_s = best["frame_s"]
    km_s = t_km / max(frames_done, 1)
    sustained = {str(r["threads"]): round((r["elapsed"] - r["t_km"]) / max(r["epochs"], 1) * 1e3, 2) for r in runs}
    if ncpu >= 64:
        time_team(64)                                    # reported, after the fact (never the fastest on this class of host: 30 ms per epoch)
        team(threads)
    all_cores = None
    if ncpu > max(team_points):
        per_all, what = _all_threads_probe(torch.tensor(mats0, dtype=torch.float32), y1, cl0, ncpu)
        all_cores = {"threads": ncpu, "ms_per_epoch": None if per_all is None else round(per_all * 1e3, 2),
                     "value": None if per_all is None else 1.0 / (2 * EPOCHS * per_all + km_s),
                     "sample": what + ("" if per_all is None else ", extrapolated to 600 epochs + the measured resample")}
        team(threads)
    cpu_model = ""
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    by = {str(k): round(v * 1e3, 2) for k, v in sorted(team_points.items())}
    if all_cores is not None and all_cores["ms_per_epoch"] is not None:
        by[str(ncpu)] = all_cores["ms_per_epoch"]
    return {"value": 1.0 / frame_s, "unit": "frames/s", "cores": threads, "kind": "port",
            "one_thread": {"value": 1.0 / (2 * EPOCHS * team_points[1] + km_s), "ms_per_epoch": round(team_points[1] * 1e3, 2),
                           "sample": "6 epochs, extrapolated to 600 + the measured resample"},
            "all_cores": all_cores,
            "ms_per_epoch_by_threads": by,
            "ms_per_epoch_sustained_full_frames": sustained,
            "cores_note": f"`value` = full registered frames at {threads} threads: the faster of the two fastest teams by 3-epoch sample (1 / 8 / 16 / 32 timed "
                          f"first, 64 after), each run through whole frames ({sustained} ms per epoch sustained -- a short sample does not predict a 600-epoch "
                          "frame); the all-threads point runs in a child process under a 3 s cap; the port's OpenMP C search is also faster than "
                          "pytorch3d's single-threaded knn_cpu, so the GPU / CPU ratio is conservative",
            "host": {"cpu_model": cpu_model, "os_cpu_count": os.cpu_count()},
            "sample": f"{frames_done} full registered frame(s) of sequence 0 (N={n_points}, K={k_clusters}, hidden {HIDDEN}): {epochs_done} Adam "
                      f"epochs at {per_epoch * 1e3:.2f} ms/epoch + {frames_done} resample_cluster at {km_s * 1e3:.1f} ms, "
                      f"{elapsed:.1f} s of host time, nothing extrapolated; oracle = torch-CPU MLP/Adam + OpenMP C L1-NN "
                      "(oracle/creg_oracle.c) + sklearn-equivalent Lloyd"}


# ------------------------------------------------------------------------------------------ ICP-style second line
def icp_variant(frames64, mats0, clusters0, dev, warm_rounds, timed_rounds):
    """SURVEY 8(d) second line: the ICP-style frame of the north star, no MLP -- K3 transform -> K4 masked
    per-cluster point-to-point ICP -> K5 dual quaternions -> K2 Lloyd k-means + change of frame -- on the
    same synthetic sequences, frames resident in HBM, sequences in lock-step."""
    from autourdf_amd import ops
    from autourdf_amd.engine import BatchIcpRegistrar
    S = len(frames64)
    breg = BatchIcpRegistrar(mats0, clusters0, S, dev)
    regs = breg.regs
    for f in range(warm_rounds):
        breg.step([fr[f] for fr in frames64])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = []
    for f in range(warm_rounds, warm_rounds + timed_rounds):
        if f == warm_rounds + timed_rounds - 1:       # state entering the last round (step() replaces, never mutates)
            saved = [(r.local, r.off, r.M) for r in regs]
        iters += [o[2] for o in breg.step([fr[f] for fr in frames64])]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_frames = timed_rounds * S
    # dominant kernel, event-timed on the stream it runs on (ops launch on torch's current stream): the
    # batched ICP launch of the last timed round, replayed from its pre-step state (which it does not modify)
    last = [fr[warm_rounds + timed_rounds - 1] for fr in frames64]
    worlds = [ops.cluster_transform(lo.to(torch.float32), of, M.to(torch.float32)) for lo, of, M in saved]
    probs = [(lo, w, of, f, M) for (lo, of, M), w, f in zip(saved, worlds, last)]
    # (one bracket around ten launches gave 762 us on one box and 1463 us on another for the same five problems: the
    #  MEDIAN of 24 individually bracketed launches after three warm-up launches, with the spread beside it)
    reps = 24
    for _ in range(3):
        ops.masked_icp_batch(probs)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    marks[0].record()
    for i in range(reps):
        ops.masked_icp_batch(probs)
        marks[i + 1].record()
    torch.cuda.synchronize()
    each = sorted(marks[i].elapsed_time(marks[i + 1]) * 1e3 for i in range(reps))
    icp_us = each[reps // 2]
    r, fr = regs[0], last[0]
    n, nf, k = r.local.shape[0], fr.shape[0], r.off.shape[0] - 1
    it = float(torch.stack(iters).double().mean())
    # algorithmic bytes of one masked-ICP launch: read local f64 + world f32 + the frame once per cluster
    # (mask scan) + poses, write world f64 + poses
    alg = S * (24 * n + 12 * n + 24 * nf * k + 128 * k + 24 * n + 128 * k)
    return {"metric": "ICP-style registered frames/sec (K3 transform + K4 masked ICP + K5 DQ + K2 k-means resample)",
            "value": round(n_frames / dt, 2), "unit": "frames/s", "ms_per_frame": round(dt / n_frames * 1e3, 3),
            "frames_timed": n_frames, "mean_icp_iterations_per_cluster": round(it, 2),
            "roofline": {"bound": "hbm", "kernel": "k_masked_icp", "avg_launch_us": round(icp_us, 1),
                         "launch_us_min_median_max": [round(each[0], 1), round(icp_us, 1), round(each[-1], 1)], "launches_timed": reps,
                         "achieved": round(alg / (icp_us * 1e-6) / 1e9, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(alg / (icp_us * 1e-6) / (HBM_PEAK_GBPS * 1e9), 6), "traffic": None,
                         "problems_per_launch": S,
                         "note": "each cluster's ICP is a serial chain of ~20-60 iterations (nearest neighbours + closed-form fit); "
                                 "latency-bound by construction (SURVEY 8(d): report honestly); algorithmic bytes = one pass over the "
                                 "cluster points, the mask scan and the poses"}}


# ------------------------------------------------------------------------------------------ stand-in registrar (CPU plumbing test)
class _StubRegistrar:
    """CREG_BENCH_STUB=1: the rank / sharding / gather logic of this file on CPU tensors with gloo -- a registrar whose
    'registration' is a fixed function of (its sequence slot's state, the frame), so poses are comparable between
    world sizes.  No kernel is involved and nothing is timed meaningfully."""

    class _Seq:
        pass

    def __init__(self, mats0, clusters0, n_tgt, n_sequences, *a, **k):
        self.S = n_sequences
        self.seqs = []
        for _ in range(n_sequences):
            r = self._Seq()
            r.m = torch.as_tensor(mats0, dtype=torch.float32).clone()
            self.seqs.append(r)
        self.plan = None
        self.last_epochs = None

    def step(self, frames64, frames32=None):
        out = []
        for r, f in zip(self.seqs, frames64):
            m2 = r.m.clone()
            m2[:, :3, 3] += 0.5 * (f.mean(0).to(torch.float32) - m2[:, :3, 3].mean(0))
            r.m = m2
            out.append((m2, torch.tensor([float(m2[:, :3, 3].abs().sum()), EPOCHS, 1e-4, 0.0])))
        return out


class _StubIcpRegistrar:
    """CREG_BENCH_STUB=1 stand-in of engine.IcpRegistrar for run_c5's rank / sharding / gather logic (no kernel)."""

    def __init__(self, mats0, clusters0, dev):
        self.M = torch.as_tensor(np.asarray(mats0), dtype=torch.float64)
        self.local, self.off = None, None

    def step(self, f):
        M = self.M.clone()
        M[:, :3, 3] += 0.5 * (f.mean(0) - M[:, :3, 3].mean(0))
        self.M = M
        return M, None, torch.tensor([3])


class _Mark:
    """A timing mark on torch's current stream (an event), or wall clock under the CPU stub."""

    def __init__(self):
        self.ev = None if STUB else torch.cuda.Event(enable_timing=True)
        self.t = 0.0

    def record(self):
        if self.ev is not None:
            self.ev.record()
        else:
            self.t = time.perf_counter()

    def elapsed_time(self, other):
        return self.ev.elapsed_time(other.ev) if self.ev is not None else (other.t - self.t) * 1e3


def frames_of(robot, sid, n_frames, n_points):
    """The frames of sequence `sid`: synthetic capsule chains (autourdf_amd.synthetic.make_sequence) or, for a `*_real` workload, the
    committed frames of the real robot (2 sequences x 10 frames; sequence ids wrap)."""
    if robot.endswith("_real"):
        d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", f"frames_{robot}.npz"))
        f = d["frames"]
        if n_frames > f.shape[1] or n_points != f.shape[2]:
            raise SystemExit(f"--workload {robot}: the fixture holds {f.shape[0]} sequences x {f.shape[1]} frames x {f.shape[2]} points "
                             f"(asked for {n_frames} frames of {n_points}): e.g. --sequences 2 --steps 12 --warmup 4")
        return [f[sid % f.shape[0], i].astype(np.float64) for i in range(n_frames)]
    from autourdf_amd.synthetic import make_sequence
    return make_sequence(robot, sid, n_frames, n_points)


def _registrar_cls():
    if STUB:
        return _StubRegistrar
    from autourdf_amd.engine import BatchRegistrar
    return BatchRegistrar


# ------------------------------------------------------------------------------------------ replay items
def capture_items(reg, frames64, frames32, n_rounds, clone_params):
    """A sequential pass over the sequences (untimed), snapshotting the state ENTERING every registration:
    item = (poses_t, clusters_t + offsets, clusters_0 + offsets, frame_t+1 (f64, f32), parameter copies of both models).
    Items are independent of each other by construction -- that is the replay / independent-frame mode."""
    items = []
    for f in range(n_rounds):
        for s, r in enumerate(reg.seqs):
            it = {"m": r.m.clone(), "f64": frames64[s][f], "f32": frames32[s][f], "seq": s, "frame": f + 1}
            if clone_params:
                it.update(pts=r.pts.clone(), off=r.off.clone(), pts_init=r.pts_init, off_init=r.off_init,
                          p_step=[p.clone() for p in r.p_step], p_anchor=[p.clone() for p in r.p_anchor])
            items.append(it)
        reg.step([frames64[s][f] for s in range(reg.S)], [frames32[s][f] for s in range(reg.S)])
    return items


def load_items(reg, batch):
    """Put a batch of items into the registrar's S slots (zero-copy: the slot's tensors ARE the item's)."""
    for r, it in zip(reg.seqs, batch):
        r.m = it["m"]
        if "pts" in it:
            r.pts, r.off, r.pts_init, r.off_init = it["pts"], it["off"], it["pts_init"], it["off_init"]
            r.p_step, r.p_anchor = it["p_step"], it["p_anchor"]


def clone_item(it):
    c = dict(it)
    c["m"] = it["m"].clone()
    if "pts" in it:
        c["p_step"] = [p.clone() for p in it["p_step"]]
        c["p_anchor"] = [p.clone() for p in it["p_anchor"]]
    return c


# ------------------------------------------------------------------------------------------ parity summary (BASELINE.md 2 "Reported")
def parity_block(state0, frame64, frame32, k_clusters, dev, rot="q", hidden=HIDDEN):
    """OUTSIDE the timed region, rank 0: the HIP path against the oracle on the first timed frame of sequence 0, from the state the
    sequence ENTERED that frame with (`state0` = poses, local clusters + offsets, a copy of the "Step" model's parameters):
      nn_idx_exact             every L1 nearest-neighbour index AND distance of both directions (predicted cloud <-> frame) bit-equal
      labels_exact_fraction    k-means labels of the frame (seeded at the entering translations) equal to the oracle's
      max_abs_dpose_8_epochs   largest |entry| difference of the [R|t] blocks train() returns after 8 epochs (inside the measured
                               divergence horizon: the north star's 1e-5 is a statement about this regime)."""
    from autourdf_amd import ops
    from oracle import chamfer, kmeans as okm, models as omodels, registration as oreg
    m, pts, off, p_step = state0
    x = ops.cluster_transform(pts, off, m)
    dx, ix, dy, iy = ops.nn_l1_bidir(x, frame32)
    xn, yn = x.cpu().numpy(), frame32.cpu().numpy()
    odx, oix = chamfer.nn_l1(xn, yn)
    ody, oiy = chamfer.nn_l1(yn, xn)
    nn_exact = bool((ix.cpu().numpy() == oix).all() and (iy.cpu().numpy() == oiy).all() and (dx.cpu().numpy() == odx).all() and (dy.cpu().numpy() == ody).all())
    seeds = m[:, :3, 3].to(torch.float64).contiguous()
    _, lab, _, n_it = ops.kmeans_lloyd(frame64, seeds)
    _, olab, _, on_it = okm.k_means(frame64.cpu().numpy(), seeds.cpu().numpy())
    lab_frac = float((lab.cpu().numpy() == olab).mean())
    order = ops.DQ_PARAM_ORDER if rot == "dq" else ops.Q_PARAM_ORDER
    params = [q.clone() for q in p_step]
    plan = ops.TrainPlan(rot, k_clusters, hidden, pts.shape[0], frame32.shape[0], epochs=8, use_graph=True, device=dev)
    bm, _, res, _, _ = plan.run(m, frame32, pts, off, params, lr=2e-4)
    omodel = omodels.DQRegMLP(hidden) if rot == "dq" else omodels.QRegMLP(True, hidden)
    omodel.load_state_dict({name: q.detach().cpu().clone() for name, q in zip(order, p_step)})
    offh = off.cpu().numpy()
    cl = [pts[offh[i]:offh[i + 1]].cpu() for i in range(len(offh) - 1)]
    _, obest, omin, _ = oreg.train(m.cpu(), frame32.cpu(), omodel, cl, rot=rot, epochs=8)
    dpose = float((bm.cpu() - obest.detach())[:, :3, :].abs().max())
    return {"nn_idx_exact": nn_exact, "labels_exact_fraction": lab_frac, "kmeans_iterations": [int(n_it), int(on_it)],
            "max_abs_dpose_8_epochs": dpose, "min_loss_rel_8_epochs": abs(float(res[0]) - omin) / abs(omin),
            "against": "oracle/ (CPU restatement of the reference path) on the first timed frame of sequence 0, outside the timed region",
            "bars": {"nn_idx_exact": True, "labels_exact_fraction": 1.0, "max_abs_dpose_8_epochs": 1e-5}}


# ------------------------------------------------------------------------------------------ roofline block
CHIP_SIMDS, CHIP_CLOCK_HZ = 256 * 4, 2.4e9              # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz max clock


def epoch_kernel_model(k_clusters, n_points, info):
    """Algorithmic HBM bytes per PROBLEM and launch of the five kernels of an epoch (head, nn, gradc, bd, l2) (DESIGN.md section 4 has the derivation),
    QRegMLP(True, hidden 512): H = 512, H2 = 768, IN = 56."""
    H, H2, IN, K, N = HIDDEN, HIDDEN + HIDDEN // 2, 56, k_clusters, n_points
    w1 = H * IN + H                                     # encoder rows (k_bwd2 updates them)
    w23 = N_PARAMS - w1                                 # hidden + output rows (k_dw updates them)
    pruned = info["pruned_target_search"] and info["pruned_predicted_search"]
    nbt, nbp = info["nn_boxes_per_lane"]
    if info.get("nn_queries_per_wave") == 16:            # round 5: sixteen queries per wave (frames above 4096 points)
        nn_name = f"k_nn_rows<{nbt}, {nbp}>"
    else:
        nn_name = (f"k_nn_plan<{'true' if info['pruned_predicted_search'] else 'false'}, {info['nn_points_per_lane']}, {nbt}, "
                   f"{nbp if info['pruned_predicted_search'] else 2}>" if info["pruned_target_search"] else "k_nn_l1<4>")
    return {
        # k_bd = two independent roles in one launch.  D: hidden + output rows, parameters + both Adam moments read and written
        # (24 B each), current activations, gradients.  B: W2 read once (its columns), the encoder rows + moments read and written,
        # g_h2, current / next encoder activation
        "bd": {"name": f"k_bd<{HIDDEN // 64}, {12 * (HIDDEN // 64)}, false>", "bound": "hbm",
               "bytes": 24 * w23 + 4 * (K * H + 2 * K * H2 + 16 * K) + 4 * H2 * H + 24 * w1 + 4 * (K * H2 + 2 * K * H + K * IN),
               # the K-row GEMMs on the matrix cores (v_mfma_f32_16x16x4_f32): g_h2 . W2 (B role) and the weight gradients g^T . act (D role)
               "mfma_flops": 2 * K * H2 * H + 2 * K * (H2 * H + 3 * (H // 2) + 4 * H)},
        # the next hidden activation: W2 + biases read again (from the freshly written buffer), next encoder activation read, h2 written
        "l2": {"name": "k_l2<8>", "bound": "hbm", "bytes": 4 * (H2 * H + H2) + 4 * (K * H + K * H2), "mfma_flops": 2 * K * H2 * H},
        # both clouds (16 B points, block-sorted copies) read once, sign bits + integer scatter counters written
        "nn_l1": {"name": nn_name, "bound": "valu", "bytes": 2 * 16 * N + 4 * N + 16 * N, "pruned": pruned},
        # points, counters, signs and predictions of the clusters read, best cloud written when the loss improved, g_h2 rows
        "gradc": {"name": "k_gradc", "bound": "latency", "bytes": (16 + 16 + 4 + 16 + 12) * N + 4 * K * H2 + 4 * 8 * H2},
        # hidden activation + output rows read, predicted cloud (16 B), sorted copy (16 B) and zeroed counters (16 B) written
        "head": {"name": "k_head<8, false>", "bound": "latency", "bytes": 4 * K * H2 + 4 * 7 * H2 + (16 + 16 + 16 + 16) * N},
    }


def roofline_block(reg, frames32, n_points, k_clusters, workload, y_index=0):
    """Per kernel, measured live with HIP events on the plan's own launches (200 back-to-back launches on the plan's stream,
    kernel + ~1 us launch gap): algorithmic bytes / launch time against the GUIDE's HBM peak.  The top-level block is the
    kernel with the LONGEST launch in THIS run (it was hard-wired to k_dw, wrong for the franka shape).  What
    cannot be read inside this process -- HBM-side traffic and SQ counters need rocprofv3's own --pmc passes -- is replayed
    from the committed summary of this workload (the newest profiles/rNN_pmc.json) and tagged with its source and with `traffic_stale`
    (the summary stores a fingerprint of the kernel sources it was collected from); `frac` of a VALU-bound
    kernel is a measured utilisation (wave-cycles the VALUs were issuing / SIMD-cycles of the launch), never an
    exhaustive-search-equivalent rate."""
    r = reg.seqs[0]
    # (every problem of the launch is staged from the same inputs and advanced 150 epochs first -- the middle of a 300-epoch train, before
    #  an early stop can set in: the nearest-neighbour launch gets shorter as the clouds align, 14.2 us after 100 epochs against 9.9 us
    #  averaged over whole trains in the rocprofv3 trace of the timed region.  The target is the frame BEFORE the one the sequence was
    #  last registered to: one frame of motion, like every train of the timed region -- until the end of round 3 it was frame 0, a jump
    #  across the whole sequence)
    prof = reg.plan.profile(r.m, frames32[0][y_index], r.pts_init, r.off_init, r.p_anchor, n_epochs=150)
    b2b = {k[:-13]: prof.pop(k) for k in [k for k in prof if k.endswith("_back_to_back")]}
    nz = prof.pop("nn_l1_problems_per_launch")
    model = epoch_kernel_model(k_clusters, n_points, reg.plan.info)
    here = os.path.dirname(os.path.abspath(__file__))
    pmc, pmc_src, pmc_stale = {}, None, None
    import glob
    cands = sorted(glob.glob(os.path.join(here, "profiles", "r[0-9][0-9]_pmc.json")))
    if cands:
        path = cands[-1]                                   # the newest round's collection
        allp = json.load(open(path))
        if workload in allp:
            pmc, pmc_src = allp[workload], f"profiles/{os.path.basename(path)}[{workload!r}]"
            from autourdf_amd.build import kernel_source_sha256
            # counters of ANOTHER build of the kernels describe other code: say so instead of dividing old cycles by new times silently
            pmc_stale = allp.get("kernel_source_sha256") != kernel_source_sha256()
    kernels = {}
    for key, m in model.items():
        us = b2b[key]
        pk = pmc.get("kernels", {}).get(key, {})
        per = pmc.get("problems_per_launch_avg", 1.0)
        e = {"kernel": m["name"], "bound": m["bound"], "avg_launch_us": round(us, 3), "problems_per_launch": nz,
             "algorithmic_bytes": m["bytes"] * nz, "achieved_GBps": round(m["bytes"] * nz / (us * 1e-6) / 1e9, 1),
             "hbm_frac": round(m["bytes"] * nz / (us * 1e-6) / (HBM_PEAK_GBPS * 1e9), 4)}
        if "mfma_flops" in m:
            # algorithmic flops of the kernel's GEMMs (without the padding of the 20 pose rows to MFMA tiles) against the DENSE f32 matrix peak
            e.update({"mfma_algorithmic_TFLOPs": round(m["mfma_flops"] * nz / (us * 1e-6) / 1e12, 2), "mfma_f32_peak_TFLOPs": 157.3,
                      "mfma_frac": round(m["mfma_flops"] * nz / (us * 1e-6) / 157.3e12, 4)})
            if "matrix_pipe_utilisation" in pk:
                e.update({"matrix_pipe_utilisation": round(pk["matrix_pipe_utilisation"], 4), "mfma_insts_per_wave": round(pk["mfma_insts_per_wave"], 1)})
        if pk:
            traffic = (2 * pk["FETCH_SIZE_KB"] + pk["WRITE_SIZE_KB"]) * 1024 / per * nz
            e.update({"traffic": round(traffic), "traffic_source": pmc_src,
                      "wasted_traffic_ratio": round(traffic / (m["bytes"] * nz), 2),
                      # VALU issue cycles of all waves of a launch / SIMD-cycles the chip offers in the launch's duration
                      "valu_utilisation": round(pk["waves"] / per * nz * pk["active_valu_cycles_per_wave"] / (us * 1e-6 * CHIP_SIMDS * CHIP_CLOCK_HZ), 4),
                      "valu_insts_per_wave": round(pk["valu_insts_per_wave"]), "utilisation_source": pmc_src})
        kernels[key] = e
    top_key = max(kernels, key=lambda k: kernels[k]["avg_launch_us"])
    top = kernels[top_key]
    if top["bound"] == "valu" and "valu_utilisation" in top:
        peak_ginst = CHIP_SIMDS * CHIP_CLOCK_HZ / 4 / 1e9          # one wave64 VALU instruction per SIMD every 4 cycles (measured: DESIGN 4)
        roof = {"bound": "valu", "kernel": top["kernel"], "achieved": round(top["valu_utilisation"] * peak_ginst, 1), "peak": round(peak_ginst, 1),
                "unit": "G wave-instructions/s", "frac": top["valu_utilisation"], "traffic": top.get("traffic"),
                "frac_note": "measured VALU utilisation of the launch: SQ_ACTIVE_INST_VALU of all its waves (rocprofv3 pass, " + str(pmc_src) +
                             ") / (launch time of this run x 1024 SIMDs x 2.4 GHz); achieved / peak restate it as wave-instructions per second"}
    else:
        roof = {"bound": "hbm", "kernel": top["kernel"], "achieved": top["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": top["hbm_frac"], "traffic": top.get("traffic")}
    roof.update({"traffic_stale": pmc_stale,
                 "traffic_stale_note": None if not pmc_stale else "the counters were collected from a tree whose train-plan kernel sources (train_engine.hip, nn_l1.h, "
                                       "creg_dev.h, creg_common.h) differ from this one, or carry no fingerprint: traffic / utilisation figures describe THAT build "
                                       "(re-run tools/collect_profiles.sh)",
                 "traffic_source": top.get("traffic_source"), "avg_launch_us": top["avg_launch_us"], "problems_per_launch": nz,
                 "algorithmic_bytes": top["algorithmic_bytes"], "wasted_traffic_ratio": top.get("wasted_traffic_ratio"),
                 "dominant_by": "longest back-to-back launch of the five epoch kernels in this run",
                 "timing_source": "HIP events around 200 back-to-back launches of each kernel on the plan's stream, in this run (kernel + ~1 us "
                                  "launch gap); launches carry the problems of the larger chain, as in the timed region, every one staged from the "
                                  "same inputs and advanced 150 epochs (mid-train) before the launches are timed",
                 "kernels": kernels,
                 "epoch_kernels_event_bracketed_us": {k: round(v, 2) if isinstance(v, float) else v for k, v in prof.items()},
                 "note": "event-bracketed per-kernel times carry ~7 us of event overhead each (upper bounds); rocprofv3 stats of the same "
                         "command are under profiles/"})
    return roof


def frame_level(roof, frames_per_s_per_gpu, n_points, k_clusters):
    """SURVEY 8(d) "Frame-level": the algorithmic HBM bytes of ONE registered frame (600 epochs x the five epoch kernels' bytes per
    problem, + the frame's k-means / change of frame: the fp64 frame read per Lloyd iteration is L2-resident and counted once) x the
    frames per second one GPU registered, against the HBM peak."""
    per_problem = sum(k["algorithmic_bytes"] / max(k["problems_per_launch"], 1) for k in roof["kernels"].values())
    resample = (24 + 4 + 24 + 12) * n_points + 2 * 128 * k_clusters          # frame read, labels, local clusters written (f64 + f32), poses
    per_frame = 2 * EPOCHS * per_problem + resample
    gbps = per_frame * frames_per_s_per_gpu / 1e9
    return {"bytes_per_frame": int(per_frame), "bytes_per_epoch_per_problem": int(per_problem), "GBps": round(gbps, 1), "peak": HBM_PEAK_GBPS,
            "frac": round(gbps / HBM_PEAK_GBPS, 4),
            "note": "sum of the five epoch kernels' algorithmic bytes per problem x 600 epochs + the resample's one pass, x frames/s of one GPU; the "
                    "epoch chain is latency-bound (five dependent launches per epoch), so this fraction moves with the chain's length, not with any one kernel"}


# ------------------------------------------------------------------------------------------ configs[4]: N=262144, K=128
def run_c5(args, ctx, robot, n_points, k_clusters, wl_tag):
    """BASELINE configs[4] (synthetic N=262144, K=128, independent frames sharded over the GPUs): SURVEY 8(d) defines this
    line over the assign / fit kernels only -- the ICP-style frame: K4 masked per-cluster ICP from the current poses
    (mask boxes of the current clusters), K5 dual quaternions, K2 Lloyd re-segmentation seeded at the new translations +
    change of frame.  (The default path's 600 epochs of an N^2 Chamfer search are 8e13 pair evaluations per frame at
    this size.)  Work items (poses, local clusters, next frame) come from a sequential pass every rank repeats untimed;
    --steps items in total are dealt round-robin."""
    from autourdf_amd.distributed import gather_poses
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    if STUB:                                            # CPU plumbing test: a stand-in registrar on a small cloud (tests/test_bench_cpu.py)
        IcpRegistrar, n_points, k_clusters = _StubIcpRegistrar, 1024, 6
    else:
        from autourdf_amd import ops
        from autourdf_amd.engine import IcpRegistrar
    world, rank, dev, dist = ctx.world, ctx.rank, ctx.dev, ctx.dist
    total = args.steps + args.warmup
    t_gen = time.perf_counter()
    seq = make_sequence(robot, 0, total + 1, n_points)
    mats0, clusters0, _ = initial_segmentation(seq[0], k_clusters, seed=0, iters=8)
    t_gen = time.perf_counter() - t_gen
    frames = [torch.as_tensor(f, dtype=torch.float64, device=dev) for f in seq[1:]]
    cap = IcpRegistrar(mats0, clusters0, dev)
    items = []
    sync = (lambda: None) if STUB else torch.cuda.synchronize
    for f in frames:                                               # the sequential pass (untimed): state entering every frame
        items.append((cap.M, cap.local, cap.off, f))
        cap.step(f)
    sync()
    job = items[args.warmup:]
    mine = job[rank::world]
    worker = IcpRegistrar(mats0, clusters0, dev)

    def register(it):
        worker.M, worker.local, worker.off = it[0], it[1], it[2]
        return worker.step(it[3])

    for it in items[:args.warmup]:
        register(it)
    sync()
    if dist is not None:
        dist.barrier()
        sync()
    poses = torch.zeros(max(len(mine), 1), k_clusters, 4, 4, dtype=torch.float64, device=dev)
    iters = []
    marks = [_Mark() for _ in range(len(mine) + 1)]     # per-frame times without a sync in the loop
    t0 = time.perf_counter()
    marks[0].record()
    for i, it in enumerate(mine):
        M_new, _, n_it = register(it)
        poses[i].copy_(M_new)
        iters.append(n_it)
        marks[i + 1].record()
    counts = [len(job[r::world]) for r in range(world)]
    t_g = time.perf_counter()
    gathered = gather_poses(poses[:len(mine)], counts=counts if dist is not None else None)
    sync()
    gather_s = time.perf_counter() - t_g
    if dist is not None:
        dist.barrier()
        sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(gathered).all() and gathered.shape[0] == args.steps
    out = None
    if rank == 0 and (STUB or getattr(args, "brief", False)):
        # a strong-scaling leg of the default --gpus N line: value, what every rank ran, the gather -- no kernel-level legs
        out = {"metric": f"ICP-style registered frames/sec (N={n_points} pts, K={k_clusters} clusters): K4 masked ICP + K5 DQ + K2 Lloyd resample",
               "value": round(args.steps / elapsed, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{robot}-shaped synthetic frames, N={n_points}, K={k_clusters} ({wl_tag})",
                          "mode": "replay / independent-frame mode: --steps work items in TOTAL from a sequential pass, dealt round-robin to the ranks",
                          "batch_sizes_per_rank": [[1] * c for c in counts], "frames_per_rank": counts,
                          "frame_ms_rank0_median": (sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(len(mine)))[len(mine) // 2] if mine else None),
                          "host_generation_s": round(t_gen, 1)},
               "gather": {"s": round(gather_s, 6), "world": world, "backend": dist.get_backend() if dist is not None else None,
                          "payload_bytes_per_rank": int(poses[:len(mine)].numel() * 8), "ragged": len(set(counts)) > 1},
               "pose_checksum": round(float(gathered.abs().sum()), 6)}
        return out
    if rank == 0:
        # the assign kernel (K2 E-step) alone, event-timed on torch's stream (ops launch there): N x K fp64 distances
        it = job[0]
        C = it[0][:, :3, 3].contiguous()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        res = {}
        for mfma in (False, True):
            ops.kmeans_assign(it[3], C, use_mfma=mfma)
            e0.record()
            for _ in range(50):
                ops.kmeans_assign(it[3], C, use_mfma=mfma)
            e1.record()
            torch.cuda.synchronize()
            res[mfma] = e0.elapsed_time(e1) * 1e3 / 50
        us = res[False]
        alg_bytes = 28.0 * n_points                                   # 24 B/point read (fp64 xyz) + 4 B label written
        flops = 8.0 * n_points * k_clusters                           # 3 fma + 1 compare-select per (point, centre) ~ 8 flops
        world32 = ops.cluster_transform(it[1].to(torch.float32), it[2], it[0].to(torch.float32))
        ops.masked_icp(it[1], world32, it[2], it[3], it[0])
        e0.record()
        for _ in range(3):
            ops.masked_icp(it[1], world32, it[2], it[3], it[0])
        e1.record()
        torch.cuda.synchronize()
        icp_us = e0.elapsed_time(e1) * 1e3 / 3
        # the kernel that OWNS the frame (rocprofv3: k_icp_nn ~59 % of a frame, profiles/): every one of its launches bracketed by HIP
        # events on the launch stream (creg_icp_nn_counters, timing on) over three more registrations of the same item, with the kernel's
        # own work counters (float32 screen trips, fp64 trips, live sources) -- VERDICT r5 item 4(a)
        ops.icp_nn_counters(reset=True, timing=True)
        n_reg = 3
        for _ in range(n_reg):
            ops.masked_icp(it[1], world32, it[2], it[3], it[0])
        torch.cuda.synchronize()
        ct = ops.icp_nn_counters(reset=True, timing=False)
        nn_launches = max(ct["nn_launches_timed"], 1.0)
        nn_us = ct["nn_launch_us_total"] / nn_launches
        pairs32, pairs64 = 512.0 * ct["f32_trips"], 512.0 * ct["f64_trips"]
        src_it = max(ct["source_iterations"], 1.0)
        nn_t = ct["nn_launch_us_total"] * 1e-6
        nn_alg = 24.0 * src_it / nn_launches                          # 24 B per live source and iteration
        nn_staged = 16.0 * 8 * 4 * ct["f32_trips"] / nn_launches      # targets staged: 16 B each, 8 per trip and lane group, 4 lane groups per wave
        ops.kmeans_lloyd(it[3], C)
        e0.record()
        _, _, _, km_it = ops.kmeans_lloyd(it[3], C)
        e1.record()
        torch.cuda.synchronize()
        km_us, km_it = e0.elapsed_time(e1) * 1e3, int(km_it)
        out = {"metric": f"ICP-style registered frames/sec (N={n_points} pts, K={k_clusters} clusters): K4 masked ICP + K5 DQ + K2 Lloyd resample",
               "value": round(args.steps / elapsed, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3 / 1.0, 3), "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{robot}-shaped synthetic frames, N={n_points}, K={k_clusters} ({wl_tag})", "n_points": n_points,
                          "k_clusters": k_clusters,
                          "mode": "replay / independent-frame mode: --steps work items in TOTAL from a sequential pass, dealt round-robin to the ranks",
                          "mean_icp_iterations_per_cluster": round(float(torch.stack(iters).double().mean()), 1),
                          "max_icp_iterations_per_frame": [int(x.max()) for x in iters],
                          "largest_cluster_points_per_frame": [int((it[2][1:] - it[2][:-1]).max()) for it in mine],
                          "frame_ms_rank0": [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(len(mine))],
                          "host_generation_s": round(t_gen, 1)},
               # the DOMINANT kernel of this frame: the ICP search (one launch per iteration over the live 64-source chunks)
               "roofline": {"bound": "latency (VALU roof and HBM roof both given)", "kernel": "k_icp_nn",
                            "avg_launch_us": round(nn_us, 2), "launches_timed": int(nn_launches), "launches_per_frame": round(nn_launches / n_reg, 1),
                            "share_of_the_icp_call": round(ct["nn_launch_us_total"] / (n_reg * icp_us), 3),
                            # VALU roof: pair evaluations of the float32 screen (3 sub + 1 mul + 2 fma = 8 flop) and of the fp64 trips behind it
                            "achieved": round(8.0 * pairs32 / nn_t / 1e12, 3), "peak": 157.3, "unit": "TFLOP/s",
                            "frac": round(8.0 * pairs32 / nn_t / 157.3e12, 5),
                            "fp64_TFLOPs": round(8.0 * pairs64 / nn_t / 1e12, 4), "fp64_vector_peak_TFLOPs": 78.6,
                            "fp64_frac": round(8.0 * pairs64 / nn_t / 78.6e12, 6),
                            "pairs_per_source_and_iteration": {"float32_screen": round(pairs32 / src_it, 1), "fp64": round(pairs64 / src_it, 2),
                                                               "needed_fraction_of_scanned": 0.961,
                                                               "note": "a pair = one (source, target) distance; a trip = 8 staged targets x 64 lanes.  The search scans the "
                                                                       "rectangle of grid cells the wave's 16 sources' (x - r, x + r) squares touch, r = the distance to the "
                                                                       "previous match; the rectangle the FINAL nearest distances span holds 0.961 of those entries "
                                                                       "(profiles/r06_icp_rect_stats.log, -DCREG_ICP_RECT_STATS build): the radius is tight, the entries are what an "
                                                                       "exact cell-grid search needs on these frames (~118 of a cluster's 256 cells while it is still moving)"},
                            "fp64_trip_fraction": round(ct["f64_trips"] / max(ct["f32_trips"], 1.0), 4), "tie_rescans_per_launch": round(ct["tie_rescans"] / nn_launches, 2),
                            "live_sources_per_launch": round(src_it / nn_launches, 0),
                            "hbm": {"algorithmic_bytes_per_launch": round(nn_alg), "achieved_GBps": round(nn_alg / (nn_us * 1e-6) / 1e9, 2), "peak": HBM_PEAK_GBPS,
                                    "frac": round(nn_alg / (nn_us * 1e-6) / (HBM_PEAK_GBPS * 1e9), 5),
                                    "staged_target_bytes_per_launch": round(nn_staged),
                                    "note": "24 B per live source and iteration (SURVEY 8(d) K4); the staged targets (16 B float32 pool entries) are L2-resident re-reads"},
                            "traffic": None,
                            "timing_source": "HIP events around EVERY k_icp_nn launch on the launch stream (creg_icp_nn_counters) over three registrations of the first "
                                             "timed item, in this run; counters = the kernel's own wave-uniform tallies",
                            "k_km_assign_standalone": {"kernel": "k_km_assign (K2 E-step, VALU form; NOT on the frame's path: the Lloyd loop runs k_km_persist / k_km_assign_pruned)",
                                                       "avg_launch_us": round(us, 2), "hbm_frac": round(alg_bytes / (us * 1e-6) / (HBM_PEAK_GBPS * 1e9), 4),
                                                       "fp64_TFLOPs": round(flops / (us * 1e-6) / 1e12, 2), "fp64_frac": round(flops / (us * 1e-6) / 1e12 / 78.6, 4),
                                                       "mfma_form_avg_launch_us": round(res[True], 2)},
                            "k_km_persist": {"kernel": "k_km_persist<2> (the frame's Lloyd loop: one persistent launch, pruned E-step + exact int64 M-step)",
                                             "us_per_iteration": round(km_us / max(km_it, 1), 2), "lloyd_iterations": km_it,
                                             "algorithmic_bytes_per_iteration": int(alg_bytes),
                                             "hbm_frac": round(alg_bytes / (km_us / max(km_it, 1) * 1e-6) / (HBM_PEAK_GBPS * 1e9), 4),
                                             "fp64_frac_of_the_full_sweep": round(flops / (km_us / max(km_it, 1) * 1e-6) / 1e12 / 78.6, 4),
                                             "note": "the pruned E-step evaluates only the centres that can be nearest per box, so the full sweep's flops over its time "
                                                     "overstate its arithmetic rate: the loop is bound by its in-launch hand-offs (wait_any 0.89 of wave-cycles, profiles/)"},
                            "kernels": {"k_masked_icp": {"avg_launch_us": round(icp_us, 1), "note": "all K clusters of one frame, whole ICP loop"},
                                        "k_means (creg_kmeans_lloyd_f64, pruned E-step + persistent kernel)":
                                            {"call_us": round(km_us, 1), "lloyd_iterations": km_it, "us_per_iteration_incl_setup": round(km_us / max(km_it, 1), 2),
                                             "note": "the Lloyd loop evaluates only the centres that can be nearest per workgroup / wave box (labels "
                                                     "identical to the full sweep); the full N x K sweep above is the standalone assign entry point"}},
                            "note": "the search is latency-bound: ~5 dependent memory round trips per wave (state, source + previous match, cell table, staged targets) at "
                                    "3 waves per SIMD (167 VGPRs), VALU active in ~0.2 of its wave-cycles and waiting on memory in ~0.6 (profiles/rNN_c5_pmc_summary.txt); "
                                    "both roofs are therefore far away by construction and are reported as what they are"},
               "pose_checksum": round(float(gathered.abs().sum()), 6)}
    return out



# ------------------------------------------------------------------------------------------ multi-GPU launch + replay batching
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(gpus, argv):
    """`python bench.py --gpus N` started WITHOUT a launcher (no RANK in the environment): run this very command line under
    `torch.distributed.run`, one rank per GPU on 127.0.0.1, and return its exit code -- the driver's own N > 1 invocation
    (which already goes through torch.distributed.run and sets RANK) never gets here."""
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this driver (RCCL needs it)
    return subprocess.call(cmd, env=env)


REPLAY_MAX_BATCH = 8          # most problems per launch of a replay round (8 in flight: 198.7 frames/s on one GPU against 157 at 5)


def replay_batches(n_items, max_batch=REPLAY_MAX_BATCH):
    """Batch sizes of one rank's replay rounds: the fewest rounds of at most `max_batch` problems, sizes differing by at
    most one (larger first), NO padding -- 7 items: [7]; 6: [6]; 50: [8, 7, 7, 7, 7, 7, 7]; 0: [].  (Round 3 padded every
    rank's items to a multiple of --sequences: 50 frames over 8 ranks were two rounds of 5 per rank instead of one of 6-7.)"""
    if n_items <= 0:
        return []
    rounds = (n_items + max_batch - 1) // max_batch
    lo, extra = divmod(n_items, rounds)
    return [lo + 1] * extra + [lo] * (rounds - extra)


# ------------------------------------------------------------------------------------------ the world (ranks, device, RCCL)
class _World:
    """world / rank / device / `torch.distributed` module (None without a process group) of this process."""

    def __init__(self, world, rank, dev, dist, note=None):
        self.world, self.rank, self.dev, self.dist, self.note = world, rank, dev, dist, note


def init_world(args):
    """One process per GPU.  Under a launcher (RANK set): the launcher's world over RCCL (gloo for the CPU plumbing test).  Started
    plainly with --gpus 1: STILL a process group -- world size 1, backend nccl -- so that the one collective of the job, the final
    all_gather of the poses (distributed.gather_poses), goes through RCCL on hardware in every single-GPU run too.
    If RCCL cannot be initialised on this box, the run continues without a group and says so."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if STUB:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no GPU visible; there is no CPU path to time)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (start it as `python bench.py --gpus N`, which launches "
                         "the ranks itself, or under torch.distributed.run with --nproc-per-node equal to --gpus)")
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run: one process per GPU over RCCL
        if STUB:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        return _World(world, rank, dev, dist, "launcher-provided process group")
    if STUB or args.no_rccl_world1:
        return _World(1, 0, dev, None, "no process group (single process)")
    try:
        dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
        t = torch.zeros(1, device=dev)
        dist.all_reduce(t)                          # builds the communicator now, not inside a timed region
        torch.cuda.synchronize()
        return _World(1, 0, dev, dist, "world-1 RCCL group created by bench.py itself (no launcher)")
    except Exception as e:                          # pragma: no cover -- a box without a usable RCCL
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass
        return _World(1, 0, dev, None, f"world-1 RCCL group could NOT be created ({type(e).__name__}: {str(e)[:200]}); ran without a process group")


def rccl_gather_probe(ctx, poses, reps=20):
    """The job's one collective on its own: `distributed.gather_poses` (RCCL all_gather_into_tensor) of this rank's pose block, timed with
    events on torch's current stream (where RCCL enqueues its kernel) over `reps` back-to-back calls after two untimed ones."""
    from autourdf_amd.distributed import gather_poses
    if ctx.dist is None or STUB:
        return None
    for _ in range(2):
        g = gather_poses(poses)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g = gather_poses(poses)
    e1.record()
    torch.cuda.synchronize()
    ok = bool(torch.equal(g[ctx.rank * poses.shape[0]:(ctx.rank + 1) * poses.shape[0]], poses))
    return {"us": round(e0.elapsed_time(e1) * 1e3 / reps, 2), "payload_bytes": poses.numel() * poses.element_size(), "world": ctx.world,
            "backend": ctx.dist.get_backend(), "collective": "all_gather_into_tensor", "own_block_returned_intact": ok, "group": ctx.note}


# ------------------------------------------------------------------------------------------ main
def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["sequences", "replay"], default="sequences")
    ap.add_argument("--sequences", type=int, default=5, help="independent sequences in flight per GPU (configs[1]: 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-icp-variant", action="store_true", help="skip the ICP-style second line (SURVEY 8(d))")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity summary against the oracle (first timed frame of sequence 0, outside the timed region)")
    ap.add_argument("--force-strong-scaling", action="store_true", help="run the strong-scaling legs at world 1 too (the single-GPU divisors of the N > 1 line, and a test of that path)")
    ap.add_argument("--no-strong-scaling", action="store_true",
                    help="world > 1: skip the strong-scaling legs (BASELINE configs[3]: 50 allegro-shaped frames, configs[4]: 200 frames of N=262144 / K=128, both dealt over the ranks)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the extra legs of the default single-GPU run: BASELINE configs[2] (franka shape) and configs[4] (N=262144, K=128)")
    ap.add_argument("--no-rccl-world1", action="store_true", help="single process without a launcher: do not create the world-1 RCCL group")
    ap.add_argument("--repeats", type=int, default=5,
                    help="sequences mode: the timed region (the same --steps frames from the same state) is run this many times in all; "
                         "`value` is the FIRST run (exactly --steps steps after --warmup), `repeats` reports median / min / max")
    ap.add_argument("--r", choices=["q", "dq", "6d", "rpy"], default="q",
                    help="pose representation (mlp_reg.py:360 --r); 'q' is the reference's default and the metric's; the others skip the roofline / cpu_baseline legs")
    ap.add_argument("--eager", action="store_true", help="eager launches instead of the captured epoch graph")
    ap.add_argument("--epochs-per-graph", type=int, default=0, help="epochs captured per hipGraph (0 = the library's 50; 300 = one replay per train)")
    ap.add_argument("--graph-branches", type=int, default=0, help="how a batch shares the GPU: 0 = the library's default (chain streams: 1 / 2 / 3 chains by batch size); -n = n chain streams "
                         "(every chain a linear graph on its own stream); +n = n parallel branches inside one graph (rounds 2-3)")
    ap.add_argument("--stop", type=int, default=200,
                    help="early-stopping patience of train() (mlp_reg.py:17: stop=200, the default and the headline's); a small value "
                         "forces early stops to show what a stopped train costs")
    ap.add_argument("--seed-offset", type=int, default=0,
                    help="shift the synthetic sequence ids (other frames, other model initialisations): the headline depends on the "
                         "trajectories the trains take -- a numerically different build is compared over several offsets")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="wx200_5",
                    help="default = the configuration BASELINE.json's metric is quoted on")
    return ap


def main(argv=None):
    import sys
    ap = build_parser()
    args = ap.parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:] if argv is None else argv))
    ctx = init_world(args)
    robot, n_points, k_clusters, wl_tag = WORKLOADS[args.workload]
    if robot.endswith("_real") and (args.steps, args.warmup, args.sequences) == (ap.get_default("steps"), ap.get_default("warmup"), ap.get_default("sequences")):
        args.steps, args.warmup, args.sequences = 12, 4, 2          # what the fixture holds: 2 sequences x 9 registrations
    if args.workload == "c5":
        args.mode = "replay"
        out = run_c5(args, ctx, robot, n_points, k_clusters, wl_tag)
    else:
        out = run_registration(args, ctx)
        # The default single-GPU invocation (what the driver runs) also carries the other single-GPU BASELINE configs, each a full run of
        # its own workload in this process AFTER the headline was measured: configs[2] (franka shape) and configs[4] (N=262144, K=128).
        headline = (args.workload == "wx200_5" and args.mode == "sequences" and args.r == "q" and ctx.world == 1 and not STUB
                    and not args.no_other_workloads and not args.eager and args.graph_branches == 0 and args.sequences == 5)
        if headline and out is not None:
            out["other_workloads"] = other_workloads(ap, ctx)
        # world > 1 (the driver's `--gpus N` line): the north star's strong-scaling questions answered by the same command -- BASELINE
        # configs[3] (50 allegro-shaped frames) and configs[4] (200 frames of N=262144 / K=128) dealt over the N ranks.  EVERY rank takes part
        # (the legs hold barriers and the gather); the weak-scaling line above stays the headline.
        strong = (args.workload == "wx200_5" and args.mode == "sequences" and args.r == "q" and (ctx.world > 1 or args.force_strong_scaling) and not args.no_strong_scaling
                  and not args.eager and args.graph_branches == 0)
        if strong:
            legs = strong_scaling_legs(ap, ctx)
            if out is not None:
                out["strong_scaling"] = legs
    if ctx.rank == 0 and out is not None:
        print(json.dumps(out))
    if ctx.dist is not None:
        ctx.dist.destroy_process_group()


STRONG_LEGS = {"configs[3]": ["--mode", "replay", "--workload", "allegro", "--steps", "50", "--warmup", "5", "--no-cpu-baseline", "--no-icp-variant", "--no-roofline",
                              "--repeats", "1"],
               "configs[4]": ["--workload", "c5", "--steps", "200", "--warmup", "2"]}


def strong_scaling_legs(ap, ctx):
    """`bench.py --gpus N`, N > 1: the two strong-scaling configurations of BASELINE.json after the weak-scaling headline, on every rank:
    `--mode replay --workload allegro --steps 50` (configs[3]: 50 frames sharded across the GPUs, RCCL gather of the poses) and
    `--workload c5 --steps 200` (configs[4]: 200 frames of N=262144 / K=128).  Total work is fixed, so value(N) / value(1 GPU of the same
    leg) is the speedup the north star asks for (>= 6x at N = 8 on configs[3]); the driver's N = 1 run carries no such leg -- run
    `python bench.py --mode replay --workload allegro --steps 50` / `--workload c5 --steps 200` for the divisor (profiles/ holds one).
    A leg that fails reports its error (on every rank alike: the legs' collectives stay matched as long as all ranks fail alike)."""
    res = {}
    for name, argv in STRONG_LEGS.items():
        t0 = time.perf_counter()
        try:
            a = ap.parse_args(["--gpus", str(ctx.world)] + argv)
            a.brief = True
            robot, n_points, k_clusters, wl_tag = WORKLOADS[a.workload]
            if a.workload == "c5":
                a.mode = "replay"
                d = run_c5(a, ctx, robot, n_points, k_clusters, wl_tag)
            else:
                d = run_registration(a, ctx)
            e = None
            if d is not None:
                keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "pose_checksum", "rccl_gather", "gather")
                e = {k: d[k] for k in keep if k in d}
                e["config"] = {k: d["config"][k] for k in ("workload", "mode", "batch_sizes_per_rank", "rounds_per_rank", "frames_per_rank", "sequences_in_flight_per_gpu",
                                                           "chains", "frame_ms_rank0_median", "host_generation_s") if k in d["config"]}
                e["argv"] = " ".join(argv)
        except Exception as ex:                      # pragma: no cover
            import traceback
            e = {"error": f"{type(ex).__name__}: {ex}", "traceback_tail": traceback.format_exc()[-600:]}
        if e is not None:
            e["leg_wall_s"] = round(time.perf_counter() - t0, 1)
            res[name] = e
        if not STUB:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    return res


def other_workloads(ap, ctx):
    """BASELINE configs[2] and configs[4] on the driver's one line: the same code paths as `--workload franka` / `--workload c5`, trimmed
    to what a reader needs (value, ms per step, checksum, the run's own roofline block).  A leg that fails reports its error instead of
    taking the headline with it."""
    legs = {"franka": ["--workload", "franka", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-icp-variant", "--repeats", "1"],
            "c5": ["--workload", "c5", "--steps", "8", "--warmup", "2"]}
    res = {}
    for name, argv in legs.items():
        t0 = time.perf_counter()
        try:
            a = ap.parse_args(argv)
            robot, n_points, k_clusters, wl_tag = WORKLOADS[a.workload]
            if name == "c5":
                a.mode = "replay"
                d = run_c5(a, ctx, robot, n_points, k_clusters, wl_tag)
            else:
                d = run_registration(a, ctx)
            keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "scaling", "dtype", "pose_checksum", "roofline")
            e = {k: d[k] for k in keep if k in d}
            e["config"] = {k: d["config"][k] for k in ("workload", "mode", "epochs_per_frame", "chains", "sequences_in_flight_per_gpu",
                                                       "mean_icp_iterations_per_cluster", "frame_ms_rank0") if k in d["config"]}
            if "rccl_gather" in d:
                e["rccl_gather"] = d["rccl_gather"]
            e["argv"] = " ".join(argv)
        except Exception as ex:                      # pragma: no cover
            import traceback
            e = {"error": f"{type(ex).__name__}: {ex}", "traceback_tail": traceback.format_exc()[-600:]}
        e["leg_wall_s"] = round(time.perf_counter() - t0, 1)
        res[name] = e
        if not STUB:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    return res


def run_registration(args, ctx):
    """One run of the default path (train Step + train Anchor + resample per frame) for `args.workload`; returns the JSON dict on rank 0."""
    robot, n_points, k_clusters, wl_tag = WORKLOADS[args.workload]
    world, rank, dev, dist = ctx.world, ctx.rank, ctx.dev, ctx.dist

    from autourdf_amd.distributed import gather_poses
    from autourdf_amd.synthetic import initial_segmentation, make_sequence
    Registrar = _registrar_cls()
    sync = (lambda: None) if STUB else torch.cuda.synchronize

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
            sync()

    replay = args.mode == "replay"
    # ---- synthetic inputs, resident in HBM before the clock starts -------------------------------
    # Frames of ONE sequence are sequentially dependent (mlp_reg.py:293-378) but sequences are independent, so the S
    # sequences advance in lock-step through ONE batched plan: every launch carries S problems.
    S = min(max(1, args.sequences), args.steps)
    if not replay and args.steps % S:
        divs = [d for d in range(S - 1, 1, -1) if args.steps % d == 0]
        S = divs[0] if divs else S
    if replay:
        # the job: --steps items in total (+ --warmup untimed ones), the same on every rank count; captured from S
        # sequences registered sequentially, which every rank repeats for itself (deterministic, untimed)
        total_items = args.steps + args.warmup
        cap_rounds = (total_items + S - 1) // S
        n_frames = cap_rounds + 1
        seq_ids = [args.seed_offset + s for s in range(S)]
    else:
        warm_rounds = (args.warmup + S - 1) // S
        timed_rounds = (args.steps + S - 1) // S
        n_frames = warm_rounds + timed_rounds + 1
        seq_ids = [args.seed_offset + rank * 1000 + s for s in range(S)]
    n_gen = n_frames if robot.endswith("_real") else max(n_frames, FRAMES_PER_SEQ)
    seq0 = frames_of(robot, 0, n_gen, n_points)
    mats0, clusters0, _ = initial_segmentation(seq0[0], k_clusters, seed=0)      # shared frame-0 state (mlp_reg.py:242-253)
    seqs = [frames_of(robot, sid, n_gen, n_points) for sid in seq_ids]
    frames64 = [[torch.as_tensor(f, dtype=torch.float64, device=dev) for f in s[1:n_frames]] for s in seqs]
    frames32 = [[f.to(torch.float32) for f in s] for s in frames64]
    hidden = 3 if args.r == "rpy" else HIDDEN            # (the reference builds RegMLP(6, 3), mlp_reg.py:285)
    if args.r != "q":
        args.no_roofline = args.no_cpu_baseline = True   # both legs are stated for the default model
    use_graph = False if args.eager else (args.epochs_per_graph if args.epochs_per_graph > 1 else True)
    reg = Registrar(mats0, clusters0, n_points, S, args.r, hidden, EPOCHS, use_graph, dev,
                    seeds=seq_ids, graph_branches=args.graph_branches)
    reg.stop = args.stop
    epochs_log = []

    def note_epochs(r=None):
        le = getattr(r if r is not None else reg, "last_epochs", None)
        if le is not None:
            epochs_log.append(le)

    if replay:
        items = capture_items(reg, frames64, frames32, cap_rounds, clone_params=not STUB)[:total_items]
        job = items[args.warmup:]
        mine = job[rank::world]                                   # round-robin: no rank holds more than one item extra
        n_mine = len(mine)
        # Every rank sizes its OWN batches: the fewest rounds of <= 8 problems, no padding (configs[3]: 50 frames over 8 ranks are
        # ONE round of 7 or 6 problems per rank; on one GPU 7 rounds of 8 / 7).  A plan is built for a batch size, so a rank owns
        # one registrar per distinct size (at most two; the capture registrar serves its own size).
        sizes = replay_batches(n_mine)
        regs = {S: reg}

        def reg_for(b):
            if b not in regs:
                regs[b] = Registrar(mats0, clusters0, n_points, b, args.r, hidden, EPOCHS, use_graph, dev,
                                    seeds=list(range(b)), graph_branches=args.graph_branches)
                regs[b].stop = args.stop
            return regs[b]

        def run_batch(batch, record=None, at=0):
            rb = reg_for(len(batch))
            load_items(rb, batch)
            out = rb.step([it["f64"] for it in batch], [it["f32"] for it in batch])
            if record is not None:
                note_epochs(rb)
                for i, (m, _) in enumerate(out):
                    record[at + i].copy_(m)

        poses = torch.zeros(max(n_mine, 1), k_clusters, 4, 4, dtype=torch.float32, device=dev)
        # untimed: the --warmup items, then one round on clones per distinct batch size of the timed region (graph capture,
        # first touch of every plan's workspace)
        warm_items = [clone_item(it) for it in items[:args.warmup]]
        pos = 0
        for b in replay_batches(len(warm_items)):
            run_batch(warm_items[pos:pos + b])
            pos += b
        for b in sorted(set(sizes)):
            run_batch([clone_item(mine[i % n_mine]) for i in range(b)])
        fence()
        epochs_log.clear()
        t0 = time.perf_counter()
        pos = 0
        for b in sizes:
            run_batch(mine[pos:pos + b], poses, pos)
            pos += b
        counts = [len(job[r::world]) for r in range(world)]
        t_g = time.perf_counter()
        gathered = gather_poses(poses[:n_mine], counts=counts if dist is not None else None)
        sync()
        replay_gather = {"s": round(time.perf_counter() - t_g, 6), "world": world, "backend": dist.get_backend() if dist is not None else None,
                         "payload_bytes_per_rank": int(poses[:n_mine].numel() * 4), "ragged": len(set(counts)) > 1,
                         "note": "the job's one collective inside the timed region (host-timed: the rank's own wait for the slowest rank is in it)"}
        fence()
        elapsed = time.perf_counter() - t0
        n_counted = args.steps
        padded = 0
        rank_rounds = [replay_batches(c) for c in counts]
        if sizes:
            reg = regs[sizes[0]]                                  # the plan the roofline leg times: this rank's (largest) replay batch
    else:
        poses = torch.zeros((warm_rounds + timed_rounds) * S, k_clusters, 4, 4, dtype=torch.float32, device=dev)

        def run_rounds(lo, hi):
            for f in range(lo, hi):
                out = reg.step([frames64[s][f] for s in range(S)], [frames32[s][f] for s in range(S)])
                note_epochs()
                for s, (m, res) in enumerate(out):
                    poses[f * S + s].copy_(m)

        def snapshot():
            """The state entering the timed region: poses / clusters by reference (a step replaces them, never writes them), the two
            models' parameters -- which the plan trains in place -- as copies."""
            return [(r.m, getattr(r, "pts", None), getattr(r, "off", None), getattr(r, "local64", None), [q.clone() for q in getattr(r, "p_step", [])],
                     [q.clone() for q in getattr(r, "p_anchor", [])]) for r in reg.seqs]

        def restore(snap):
            for r, (m_, pts_, off_, l64, ps, pa) in zip(reg.seqs, snap):
                r.m = m_
                if pts_ is not None:
                    r.pts, r.off = pts_, off_
                if l64 is not None:
                    r.local64 = l64
                for dst, src in zip(getattr(r, "p_step", []), ps):
                    dst.copy_(src)
                for dst, src in zip(getattr(r, "p_anchor", []), pa):
                    dst.copy_(src)

        sync()
        run_rounds(0, warm_rounds)
        n_rep = max(1, args.repeats)
        snap = snapshot() if n_rep > 1 else None
        if not STUB and rank == 0:      # sequence 0 entering its first timed frame (the parity summary, computed after the timing)
            r0 = reg.seqs[0]
            parity_state = (r0.m, r0.pts, r0.off, [q.clone() for q in r0.p_step])
        fence()
        epochs_log.clear()
        t0 = time.perf_counter()
        run_rounds(warm_rounds, warm_rounds + timed_rounds)
        timed = poses[warm_rounds * S: warm_rounds * S + args.steps]
        gathered = gather_poses(timed)                     # the one exchange of the job (RCCL all_gather; world 1 included when a group exists)
        fence()
        elapsed = time.perf_counter() - t0
        n_counted = world * args.steps
        padded = timed_rounds * S - args.steps
        rank_rounds = [[S] * timed_rounds for _ in range(world)]
        # --repeats: the SAME timed region again (same state, same frames, same gather) -- `value` stays the first run; identical checksums
        # are a run-to-run determinism check of the whole frame loop on the side
        rep_elapsed, rep_same = [elapsed], True
        for _ in range(n_rep - 1):
            first = gathered.clone()
            restore(snap)
            keep_log = list(epochs_log)
            fence()
            tr = time.perf_counter()
            run_rounds(warm_rounds, warm_rounds + timed_rounds)
            g2 = gather_poses(poses[warm_rounds * S: warm_rounds * S + args.steps])
            fence()
            rep_elapsed.append(time.perf_counter() - tr)
            rep_same = rep_same and bool(torch.equal(g2, first))
            epochs_log[:] = keep_log
    if replay:
        rep_elapsed, rep_same = [elapsed], True
    if dist is not None:
        t = torch.tensor(rep_elapsed, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rep_elapsed = [float(v) for v in t.tolist()]
        elapsed = rep_elapsed[0]
    assert torch.isfinite(gathered).all() and gathered.shape[0] == n_counted, (gathered.shape, n_counted)
    # the job's one collective, on its own (every rank takes part)
    mine_block = poses[:max(n_mine, 1)] if replay else poses[warm_rounds * S: warm_rounds * S + args.steps]
    rccl = rccl_gather_probe(ctx, mine_block.contiguous()) if (not replay or len({len(job[r::world]) for r in range(world)}) == 1) else None

    out = None
    if rank == 0:
        ep = {}
        if epochs_log:
            e = torch.stack([torch.cat([pair[i].to(torch.float32).reshape(-1) for pair in epochs_log]) for i in (0, 1)]).cpu()   # (2, problems)
            ep = {"epochs_run_step": {"min": int(e[0].min()), "mean": round(float(e[0].mean()), 1)},
                  "epochs_run_anchor": {"min": int(e[1].min()), "mean": round(float(e[1].mean()), 1)},
                  "early_stop": bool((e < EPOCHS).any()),
                  "stop_patience": args.stop,
                  "early_stop_note": "early stopping is live (mlp_reg.py:107-111; stop=200 unless --stop says otherwise); the epochs of a "
                                     "captured graph that follow a stop return at their first barrier (each kernel reads the stop flag with "
                                     "its first loads), so a stopped train costs its launch floors only"}
        out = {"metric": f"registered frames/sec (N={n_points} pts, K={k_clusters} clusters)", "value": round(n_counted / elapsed, 4),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "strong" if replay else "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{robot}-shaped, {S} sequences x 10 frames per GPU, N={n_points}, K={k_clusters} ({wl_tag}); "
                                      f"1 step = 1 registered frame = 2 x 300 Adam epochs ({MODEL_NAMES[args.r]}, L1 Chamfer) "
                                      "+ Lloyd k-means resample", "n_points": n_points, "k_clusters": k_clusters,
                          "mode": ("replay / independent-frame mode: --steps work items in TOTAL, captured from a sequential pass, dealt "
                                   "round-robin to the ranks (SURVEY 8(e); frames of a sequence cannot be sharded otherwise)") if replay else
                                  "sequences: every rank registers its own sequences frame by frame (weak scaling)",
                          "epochs_per_frame": 2 * EPOCHS, **ep, "launch": "eager" if args.eager else "hipGraph",
                          "chains": (lambda pl: None if pl is None else {"count": pl.info["graph_branches"],
                                                                         "form": "parallel branches of one graph" if args.graph_branches > 0 else
                                                                         "chain streams: every chain a linear graph on its own stream, chain 0 on the caller's"})(getattr(reg, "plan", None)),
                          "sequences_in_flight_per_gpu": max(rank_rounds[0]) if (replay and rank_rounds[0]) else S,
                          "padded_steps_timed_not_counted": padded,
                          # what every rank ran inside the timed region: its rounds and the problems each carried
                          "rounds_per_rank": [len(rr) for rr in rank_rounds], "batch_sizes_per_rank": rank_rounds,
                          "padded_steps_per_rank": [0] * world if replay else [padded] * world,
                          "sharding": ("items round-robin over ranks" if replay else "sequences per rank") + ", final all_gather of poses"
                                      if world > 1 else "single GPU"},
               "pose_checksum": round(float(gathered.double().abs().sum()), 6)}
        if len(rep_elapsed) > 1:
            vals = sorted(n_counted / e for e in rep_elapsed)
            out["repeats"] = {"n": len(rep_elapsed), "values": [round(n_counted / e, 2) for e in rep_elapsed],
                              "median": round(vals[len(vals) // 2], 2), "min": round(vals[0], 2), "max": round(vals[-1], 2),
                              "poses_identical_across_repeats": rep_same,
                              "note": "the timed region run n times from the same state (parameters restored, same frames, same final gather); "
                                      "`value` is the first of them -- exactly --steps steps after --warmup -- the others only show the spread on this box"}
        if replay:
            out["gather"] = replay_gather
        if rccl is not None:
            out["rccl_gather"] = rccl
            if world == 1:
                out["rccl_world1_gather_us"] = rccl["us"]
        if not STUB and not args.no_roofline:
            out["roofline"] = roofline_block(reg, frames32, n_points, k_clusters, args.workload,
                                             y_index=0 if replay else max(warm_rounds + timed_rounds - 2, 0))
            out["roofline"]["frame_level"] = frame_level(out["roofline"], out["value"] / world, n_points, k_clusters)
        if not STUB and not replay and not args.no_parity:
            try:
                out["parity"] = parity_block(parity_state, frames64[0][warm_rounds], frames32[0][warm_rounds], k_clusters, dev, args.r, hidden)
            except Exception as ex:                      # pragma: no cover -- the summary must not take the measured line with it
                out["parity"] = {"error": f"{type(ex).__name__}: {ex}"}
        if not STUB and world == 1 and not args.no_icp_variant and not replay:      # a one-GPU secondary line: not while other ranks wait
            out["icp_variant"] = icp_variant(frames64, mats0, clusters0, dev, warm_rounds, timed_rounds)
        if not STUB and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(seq0, mats0, clusters0, n_points, k_clusters)
            out["speedup_vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    return out


if __name__ == "__main__":
    main()
