"""Per-sequence cluster registration on the MI355X -- drop-in for reference PointCloud/mlp_reg.py.

Same public surface (``train`` :17, ``calculate_pc`` :155, ``resample_cluster`` :172, ``match`` :240,
the CLI flags :394-404 and the module globals the ``__main__`` block sets :390-426), same files
written (``matrix/{t:04}.npy``, ``cluster/{t:04}.npz``, ``loss.txt``).  What changed is where the
work runs:

* ``train``            one device-resident plan in libcreg.so (A1): 300 epochs of pose MLP,
                       calculate_pc, L1 Chamfer forward/backward, Adam and ReduceLROnPlateau with
                       no host round trip (the reference syncs on ``loss.item()`` every epoch).
* ``calculate_pc``     HIP kernel K3 (differentiable).
* ``resample_cluster`` fp64 Lloyd k-means + stable grouping + inverse-pose change of frame, K2.
* ``masked_icp``       (``--mlp_icp``) one launch for all clusters, K4.

There is no CPU path: without libcreg.so / an MI355X every entry point raises.
Run as ``python -m autourdf_amd.mlp_reg --robot wx200_5`` from a directory holding
``parameters.json`` and ``data/raw/...`` (what scripts/registration.sh does).
"""
import argparse
import glob
import json
import os
import queue
import threading

import numpy as np
import torch

from . import _lib, ops
from .cluster_icp import PointCloud, Segments, masked_icp
from .helper_functions import load_pc_npz, save_pc_npz
from .model_utils import DQRegMLP, QRegMLP, RegMLP, RRegMLP

# module globals, set by main() exactly like the reference's __main__ block; ROT has a default so
# that `import mlp_reg; mlp_reg.train(...)` works without NameError.
ROT = "q"
NORMAL = False          # --normal (mlp_reg.py:399): normals + 6-D k-means in the re-segmentation
MLP_ICP = False
DEVICE = None
EPOCHS = 300            # mlp_reg.py:60
USE_GRAPH = True
_PLANS = {}


def _plan(rot, k, hidden, n_pred, n_tgt, device):
    key = (rot, k, hidden, n_pred, n_tgt, EPOCHS, USE_GRAPH, str(device))
    if key not in _PLANS:
        _PLANS[key] = ops.TrainPlan(rot, k, hidden, n_pred, n_tgt, epochs=EPOCHS, use_graph=USE_GRAPH, device=device)
    return _PLANS[key]


def _model_params(model):
    if isinstance(model, QRegMLP):
        rot, order = "q", ops.Q_PARAM_ORDER
    elif isinstance(model, DQRegMLP):
        rot, order = "dq", ops.DQ_PARAM_ORDER
    elif isinstance(model, RRegMLP):
        rot, order = "6d", ops.Q_PARAM_ORDER
    elif isinstance(model, RegMLP):
        if not model.multi_decoder:
            raise NotImplementedError("train(): the single-decoder RegMLP is never constructed on the reference path (mlp_reg.py:285)")
        rot, order = "rpy", ops.Q_PARAM_ORDER
    else:
        raise NotImplementedError(f"train(): no HIP plan for {type(model).__name__}")
    named = dict(model.named_parameters())
    params = [named[n].data for n in order]
    return rot, params, model.encoder[0].out_features


def train(m, y, model, clusters, stop=200, learning_rate=0.0002, scheduler_patience=5, scheduler_factor=0.7):
    """The reference's Adam loop (mlp_reg.py:17-152) as one asynchronous device plan.

    Args as in the reference: m (K,4,4) fp32 poses, y (N,3) fp32 target cloud, model (QRegMLP / DQRegMLP / RRegMLP /
    RegMLP for --r q / dq / 6d / rpy, updated in place), clusters (list of K (M_k,3) fp32 local clouds).
    Returns (pred_pcd_np, pred_pcd, best_m, min_loss) like the reference.
    """
    rot, params, hidden = _model_params(model)
    if rot != ROT:
        raise ValueError(f"model {type(model).__name__} does not match ROT={ROT!r}")
    dev = m.device
    pts, off = ops.pack_clusters(clusters, dev)
    plan = _plan(rot, m.shape[0], hidden, pts.shape[0], y.shape[0], dev)
    best_m, best_pred, result, _, _ = plan.run(m, y, pts, off, params, lr=learning_rate, factor=scheduler_factor,
                                               patience=scheduler_patience, stop=stop)
    res = result.cpu()                                   # the one host sync of the whole loop
    min_loss, epochs_run = float(res[0]), int(res[1])
    if epochs_run < EPOCHS:
        print(f"Early stopping triggered after {epochs_run - 1} epochs")
    off_h = off.cpu().numpy()
    pred_h = best_pred.cpu().numpy()
    pred_pcd_np = [pred_h[off_h[i]:off_h[i + 1]] for i in range(len(clusters))]
    pred_pcd = [PointCloud(p) for p in pred_pcd_np]
    print("Best Loss:", min_loss)
    return pred_pcd_np, pred_pcd, best_m, min_loss


def calculate_pc(local_clusters, matrices):
    """list of (M_k,3) local clusters, (K,4,4) poses -> list of world-frame clusters (mlp_reg.py:155-170)."""
    dev = matrices.device
    pts, off = ops.pack_clusters(local_clusters, dev)
    out = ops.cluster_transform(pts, off, matrices.to(torch.float32))
    sizes = [int(c.shape[0]) for c in local_clusters]
    return list(torch.split(out, sizes, dim=0))


def resample_cluster(segments, idx, n_clusters, matrices, normal=False, visual=False):
    """Re-segment frame ``idx`` around the current poses and express each cluster in its pose frame
    (mlp_reg.py:172-237): k_means(init = pose translations, n_init=1) -> labels -> inv(M_k).[p;1]."""
    if visual:
        raise NotImplementedError("visual=True needs Open3D's GUI (out of scope)")
    dev = _lib.device(globals().get("DEVICE"))
    pc_np = np.asarray(segments.pc_list[idx].points)
    X = torch.as_tensor(pc_np, dtype=torch.float64, device=dev).contiguous()
    matrices = np.asarray(matrices)
    if matrices.shape[0] != n_clusters:
        raise ValueError("matrices must hold n_clusters poses")
    init = torch.as_tensor(matrices[:, :3, 3], device=dev).to(torch.float64).contiguous()
    if normal:
        # mlp_reg.py:190-203: normals (hybrid radius 0.1 / 30 neighbours, consistently oriented), then k_means over
        # [xyz | 0.5 n] seeded at [translation | 0]
        from .normals import point_features
        feat, nrm = point_features(pc_np)
        segments.pc_list[idx].normals = nrm                    # the reference leaves them on the point cloud too
        X6 = torch.as_tensor(feat, dtype=torch.float64, device=dev).contiguous()
        init6 = torch.cat([init, torch.zeros_like(init)], 1).contiguous()
        _, labels, _, _ = ops.kmeans_lloyd_nd(X6, init6)
    else:
        _, labels, _, _ = ops.kmeans_lloyd(X, init)
    # mlp_reg.py:211: np.linalg.inv(matrices[i]) in the poses' own dtype (float32 on the default path, float64 after
    # masked_icp) -- the very same host call, so the inverse has the reference's bits on whatever BLAS numpy carries;
    # the (N,3) change of frame itself runs on the device
    inv = np.linalg.inv(matrices).astype(np.float64)          # stacked call: LAPACK ?gesv per (4,4) matrix, the same bits
    local, off = ops.group_to_local(X, labels, torch.as_tensor(inv, device=dev).contiguous(), m_is_inverse=True)
    off_h, local_h = off.cpu().numpy(), local.cpu().numpy()
    if (np.diff(off_h) == 0).any():
        import warnings
        warnings.warn(f"Number of distinct clusters ({int((np.diff(off_h) > 0).sum())}) found smaller than "
                      f"n_clusters ({n_clusters}). Possibly due to duplicate points in X.")
    return [local_h[off_h[i]:off_h[i + 1]] for i in range(n_clusters)]


def _make_models():
    if ROT == "dq":
        print("Using DQRegMLP")
        return DQRegMLP(hidden_dim=512).to(DEVICE), DQRegMLP(hidden_dim=512).to(DEVICE)
    if ROT == "q":
        print("Using QRegMLP")
        return QRegMLP(True, hidden_dim=512).to(DEVICE), QRegMLP(True, hidden_dim=512).to(DEVICE)
    if ROT == "rpy":
        print("Using RegMLP")
        return RegMLP(6, 3).to(DEVICE), RegMLP(6, 3).to(DEVICE)          # the reference's call, mlp_reg.py:285
    print("Using RRegMLP")
    return RRegMLP(hidden_dim=512).to(DEVICE), RRegMLP(hidden_dim=512).to(DEVICE)


class _FileWriter:
    """``matrix/NNNN.npy`` / ``cluster/NNNN.npz`` (mlp_reg.py:376-378) written by a worker thread, in submission order, while the
    next frame registers: ``np.savez`` of a frame's 20 clusters is ~1.2 ms of host time, and written in line it left the GPU idle
    for a fifth of a lock-step round (5 sequences x 200 frames, files and start-up included: 6.9 -> 6.0 ms per frame end to end
    with this and the worker-side fetch of match_all, profiles/r04_cli_end_to_end.log; the registration alone is 5.5).
    Same functions, same bytes; an error in the worker is re-raised by the next ``submit`` or by ``close``.
    At most ``MAX_PENDING`` frames wait behind the worker: a job of the lock-step engine pins a frame's stacked DEVICE tensors
    (S x N x 3 float64 + poses: ~30 MB at 5 x 262144 points) until it is written, so the queue bounds device memory, not just count."""
    MAX_PENDING = 8

    def __init__(self):
        self._q = queue.Queue(maxsize=self.MAX_PENDING)
        self._err = None
        self._t = threading.Thread(target=self._run, name="creg-file-writer", daemon=True)
        self._t.start()

    def _run(self):
        try:
            if getattr(DEVICE, "index", None) is not None:      # (one process per GPU: the worker's copies go to the rank's device)
                torch.cuda.set_device(DEVICE)
        except BaseException as e:                  # noqa: BLE001
            self._err = e
        while True:                                 # keeps draining after an error, so that submit / close never block on a dead worker
            job = self._q.get()
            if job is None:
                return
            if self._err is None:
                try:
                    job[0](*job[1])
                except BaseException as e:          # noqa: BLE001  (handed to the submitting thread)
                    self._err = e

    def submit(self, fn, *args):
        if self._err is not None:
            raise self._err
        while True:
            try:
                self._q.put((fn, args), timeout=5.0)
                return
            except queue.Full:
                if not self._t.is_alive():
                    raise RuntimeError("the file-writer thread died") from self._err

    def close(self, propagating: bool = False):
        """Drain and stop.  ``propagating``: another exception is already on its way out of the frame loop -- the worker's stored
        error (usually a consequence of it) must not replace it."""
        self._q.put(None)
        self._t.join()
        if self._err is not None and not propagating:
            raise self._err


_FETCH_STREAM = {}


def _fetch_and_save(save_dirs, t, stacked, ev, losses):
    """Worker-side half of a lock-step frame: the stacked device results -> matrix/TTTT.npy, cluster/TTTT.npz per sequence, the
    best losses (appended in frame order: one worker, jobs in submission order)."""
    ev.synchronize()
    dev = stacked[0].device
    side = _FETCH_STREAM.get(dev)
    if side is None:
        side = _FETCH_STREAM[dev] = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        offs, locs, ms, rs = [x.cpu().numpy() for x in stacked]
    for s, sd in enumerate(save_dirs):
        off, local = offs[s], locs[s]
        np.save(sd + f"matrix/{t:04}.npy", ms[s])
        save_pc_npz([local[off[j]:off[j + 1]] for j in range(len(off) - 1)], sd + f"cluster/{t:04}.npz")
        losses[s].append(float(rs[s][0]))


def register_sequence(seg, step_matrices, step_cluster_np, save_dir=None, mlp_icp=False, models=None, loss_log=None):
    """Frames 1..T-1 of one sequence (the loop body of ``match``, mlp_reg.py:293-378), on arrays.
    Returns (list of (K,4,4) poses per frame incl. frame 0, best losses)."""
    K = len(step_cluster_np)
    m_t = torch.tensor(np.asarray(step_matrices), dtype=torch.float32).to(DEVICE)
    cl_t = [torch.tensor(step_cluster_np[i], dtype=torch.float32).to(DEVICE) for i in range(K)]
    cl_init = [c.clone() for c in cl_t]
    icp_src = step_cluster_np            # mlp_reg.py:248/253: assigned once; masked_icp's source for EVERY frame (:325)
    model, model_rf = models if models is not None else _make_models()
    poses, best_losses = [np.asarray(step_matrices)], []
    writer = _FileWriter() if save_dir is not None else None
    # (whether an exception of THIS frame loop is on its way out is tracked here: sys.exc_info() in a `finally` is also set when the
    #  caller merely runs inside somebody's `except` block, and a genuine writer failure was then swallowed -- ADVICE r5)
    failed = False
    try:
        _register_frames(seg, K, m_t, cl_t, cl_init, icp_src, model, model_rf, mlp_icp, save_dir, writer, poses, best_losses)
    except BaseException:
        failed = True
        raise
    finally:
        if writer is not None:
            writer.close(propagating=failed)
    return poses, best_losses


def _register_frames(seg, K, m_t, cl_t, cl_init, icp_src, model, model_rf, mlp_icp, save_dir, writer, poses, best_losses):
    for i in range(0, seg.data_size - 1):
        target_np = np.array(seg.pc_list[i + 1].points)
        target = torch.tensor(target_np, dtype=torch.float32).to(DEVICE)
        if mlp_icp:
            pred_np, _, step_m, best_loss = train(m=m_t, y=target, model=model, clusters=cl_t)
            best_losses.append(best_loss)
            step_m_np = step_m.detach().cpu().numpy()
            _, matrices = masked_icp(icp_src, pred_np, target_np, step_m_np, False, ori=False)
            new_seg_np = resample_cluster(seg, i + 1, K, matrices, NORMAL)
            m_t = torch.tensor(matrices, dtype=torch.float32).to(DEVICE)
            out_m = matrices
        else:
            _, _, step_m, _ = train(m=m_t, y=target, model=model, clusters=cl_t)                 # "Step"
            m_t = step_m.detach().clone().to(DEVICE)
            _, _, step_m, best_loss = train(m=m_t, y=target, model=model_rf, clusters=cl_init,
                                            learning_rate=0.0001)                                 # "Anchor"
            m_t = step_m.detach().clone().to(DEVICE)
            best_losses.append(best_loss)
            out_m = step_m.detach().cpu().numpy()
            new_seg_np = resample_cluster(seg, i + 1, K, out_m, NORMAL)
        cl_t = [torch.tensor(new_seg_np[j], dtype=torch.float32).to(DEVICE) for j in range(K)]
        poses.append(out_m)
        if save_dir is not None:
            writer.submit(np.save, save_dir + f"matrix/{(i + 1):04}.npy", out_m)
            writer.submit(save_pc_npz, new_seg_np, save_dir + f"cluster/{(i + 1):04}.npz")


def match(data_dir, idx):
    """One sequence ("video"): frame-0 state (fresh k-means++ for the very first sequence, otherwise
    reloaded from the first output directory, mlp_reg.py:242-253), then frames 1..T-1."""
    save_dir_list = sorted(glob.glob(f"data/part/{ROBOT}_{NUM_SEG}_seg/{STEP_SZIE}_deg_{NUM_CAMERAS}_cams/*/"))
    if len(save_dir_list) == 0:
        seg0 = Segments(RAW_PATH_LIST[0])
        seg0.k_means_cluster(0, NUM_SEG, NORMAL)
        step_matrices = np.array(seg0.init_matrix_list)
        step_cluster_np = seg0.init_segment_list
    else:
        first_dir = save_dir_list[0]
        step_matrices = np.load(first_dir + "matrix/0000.npy")
        step_cluster_np = load_pc_npz(first_dir + "cluster/0000.npz")
    sub_dir = data_dir.split("/")[-2]
    save_dir = f"data/part/{ROBOT}_{NUM_SEG}_seg/{STEP_SZIE}_deg_{NUM_CAMERAS}_cams/{sub_dir}/"
    os.makedirs(save_dir + "cluster", exist_ok=True)
    os.makedirs(save_dir + "matrix", exist_ok=True)
    np.save(save_dir + "matrix/0000.npy", step_matrices)
    save_pc_npz(step_cluster_np, save_dir + "cluster/0000.npz")
    seg = Segments(data_dir)
    _, best_losses = register_sequence(seg, step_matrices, step_cluster_np, save_dir, mlp_icp=MLP_ICP)
    if LOSS:
        np.savetxt(save_dir + "loss.txt", best_losses)


def _ensure_frame0(first_dir):
    """The frame-0 state every sequence starts from (mlp_reg.py:242-253), written once into the first sequence's
    output directory if no output exists yet."""
    base = f"data/part/{ROBOT}_{NUM_SEG}_seg/{STEP_SZIE}_deg_{NUM_CAMERAS}_cams/"
    if sorted(glob.glob(base + "*/")):
        return
    seg0 = Segments(RAW_PATH_LIST[0])
    seg0.k_means_cluster(0, NUM_SEG, NORMAL)
    sd = base + first_dir.split("/")[-2] + "/"
    os.makedirs(sd + "cluster", exist_ok=True)
    os.makedirs(sd + "matrix", exist_ok=True)
    np.save(sd + "matrix/0000.npy", np.array(seg0.init_matrix_list))
    save_pc_npz(seg0.init_segment_list, sd + "cluster/0000.npz")


def match_all(data_dirs):
    """Every sequence of a run in lock-step (not in the reference, which calls match() per sequence):
    sequences are independent once sequence 0 has produced the shared frame-0 state (mlp_reg.py:242-253), so the
    S default-path (MLP + MLP) registrations advance as ONE batched train plan per step -- the same kernels and
    arithmetic as S match() calls, the same files, ~3x the throughput.  Falls back to match() per sequence when
    the sequences differ in length or size.  --mlp_icp takes the same route with one batched
    masked-ICP launch per step (mlp_reg.py:296-332)."""
    from .engine import BatchRegistrar
    segs = [Segments(d) for d in data_dirs]
    same = len({(sg.data_size, len(sg.pc_list[0].points)) for sg in segs}) == 1 and \
        all(len(p.points) == len(segs[0].pc_list[0].points) for sg in segs for p in sg.pc_list)
    if not same or len(segs) < 1:
        for i, d in enumerate(data_dirs):
            match(d, i)
        return
    base = f"data/part/{ROBOT}_{NUM_SEG}_seg/{STEP_SZIE}_deg_{NUM_CAMERAS}_cams/"
    done = sorted(glob.glob(base + "*/"))
    if len(done) == 0:                                   # frame-0 state from the first raw sequence (mlp_reg.py:244-249)
        seg0 = Segments(RAW_PATH_LIST[0])
        seg0.k_means_cluster(0, NUM_SEG, NORMAL)
        step_matrices, step_cluster_np = np.array(seg0.init_matrix_list), seg0.init_segment_list
    else:
        step_matrices, step_cluster_np = np.load(done[0] + "matrix/0000.npy"), load_pc_npz(done[0] + "cluster/0000.npz")
    n = len(segs[0].pc_list[0].points)
    if sum(len(c) for c in step_cluster_np) != n:
        # the frame-0 state comes from another run / the first raw sequence and holds another point count than these
        # frames: "Anchor" (n0 points) and "Step" (n points) would need two plan shapes -- match() keys its plans on both
        for i, d in enumerate(data_dirs):
            match(d, i)
        return
    save_dirs = [base + d.split("/")[-2] + "/" for d in data_dirs]
    for sd in save_dirs:
        os.makedirs(sd + "cluster", exist_ok=True)
        os.makedirs(sd + "matrix", exist_ok=True)
        np.save(sd + "matrix/0000.npy", step_matrices)
        save_pc_npz(step_cluster_np, sd + "cluster/0000.npz")
    models = [_make_models() for _ in segs]
    hidden = models[0][0].encoder[0].out_features        # 512, but 3 for --r rpy (RegMLP(6, 3), mlp_reg.py:285)
    reg = BatchRegistrar(np.asarray(step_matrices, np.float32), [np.asarray(c, np.float64) for c in step_cluster_np], n,
                         len(segs), ROT, hidden, EPOCHS, USE_GRAPH, DEVICE, models=models)
    reg.normal = NORMAL                                  # --normal: the re-segmentation over [xyz | 0.5 n], per sequence, inside the lock-step round
    losses = [[] for _ in segs]
    writer = _FileWriter()
    failed = False
    try:
        for i in range(segs[0].data_size - 1):
            frames = [torch.as_tensor(np.asarray(sg.pc_list[i + 1].points), dtype=torch.float64, device=DEVICE) for sg in segs]
            out = reg.step_mlp_icp(frames) if MLP_ICP else reg.step(frames)
            # One stacked tensor per kind for all sequences (device side), an event behind them, and the rest on the worker: it
            # waits for the event ON THE HOST, copies on its own stream (a copy on the main stream would queue behind the next
            # frame's trains), slices and writes -- the main thread goes straight on to the next frame.
            stacked = (torch.stack([r.off for r in reg.seqs]), torch.stack([r.local64 for r in reg.seqs]),
                       torch.stack([o[0] for o in out]), torch.stack([o[1] for o in out]))
            ev = torch.cuda.Event()
            ev.record()
            writer.submit(_fetch_and_save, save_dirs, i + 1, stacked, ev, losses)
    except BaseException:
        failed = True
        raise
    finally:
        writer.close(propagating=failed)
    if LOSS:
        for sd, l in zip(save_dirs, losses):
            np.savetxt(sd + "loss.txt", l)


def _shard_for_rank(dirs):
    """One process per GPU (python -m torch.distributed.run ... -m autourdf_amd.mlp_reg): the sequences are dealt
    round-robin to the ranks, nothing is exchanged on the data path; only the frame-0 state every sequence starts from
    is shared -- through the file system, as the reference shares it between its sequential match() calls
    (mlp_reg.py:242-253): rank 0 writes it, a barrier, everybody reads it.  Backend: RCCL ("nccl"); the CPU
    plumbing test of tests/test_distributed_cpu.py sets CREG_DIST_BACKEND=gloo."""
    global DEVICE
    import torch.distributed as dist
    from .distributed import shard_sequences
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    backend = os.environ.get("CREG_DIST_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        DEVICE = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=DEVICE)
    else:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if rank == 0:
        _ensure_frame0(dirs[0])
    dist.barrier()
    return [dirs[i] for i in shard_sequences(len(dirs), rank, world)]


def main(argv=None):
    from . import prefer_device_kernargs
    prefer_device_kernargs()                    # (the command-line entry point: before the first device call)
    global DEVICE, ROBOT, NUM_SEG, DOF, STEP_SZIE, NUM_CAMERAS, MLP_ICP, VIS, ROT, LOSS, NORMAL, RAW_PATH_LIST
    if not torch.cuda.is_available():
        raise RuntimeError("autourdf_amd.mlp_reg needs an MI355X: no GPU is visible and there is no CPU path")
    DEVICE = _lib.device()                      # the current device, index spelled out
    print("Using device:", DEVICE)
    parser = argparse.ArgumentParser()
    parser.add_argument("--robot", type=str, default="nao")
    parser.add_argument("--mlp_icp", action="store_true")
    parser.add_argument("--visual", action="store_true")
    parser.add_argument("--loss", action="store_true")
    parser.add_argument("--normal", action="store_true")
    parser.add_argument("--num_cameras", type=int, default=20)
    parser.add_argument("--step_size", type=int, default=4)
    parser.add_argument("--num_video", type=int, default=5)
    parser.add_argument("--r", type=str, default="q", choices=["q", "rpy", "dq", "6d"])
    parser.add_argument("--sequential", action="store_true",
                        help="one match() per sequence like the reference's main loop (default: all sequences in lock-step)")
    args = parser.parse_args(argv)
    with open("parameters.json") as f:
        robot_params = json.load(f)[args.robot]
    ROBOT, NUM_SEG, DOF = args.robot, robot_params["num_seg"], robot_params["dof"]
    STEP_SZIE, NUM_CAMERAS = args.step_size, args.num_cameras
    MLP_ICP, VIS, ROT, LOSS, NORMAL = args.mlp_icp, args.visual, args.r, args.loss, args.normal
    if VIS:
        raise NotImplementedError("--visual needs Open3D's GUI (out of scope)")
    RAW_PATH_LIST = sorted(glob.glob(f"data/raw/{ROBOT}/{STEP_SZIE}_deg_{NUM_CAMERAS}_cams/*/"))
    if len(RAW_PATH_LIST) == 0:
        RAW_PATH_LIST = sorted(glob.glob(f"data/raw/{ROBOT}/*/"))
    print(f"Found {len(RAW_PATH_LIST)} raw data directories")
    dirs = RAW_PATH_LIST[: args.num_video]
    distributed = "RANK" in os.environ and bool(dirs)          # launched by torch.distributed.run (any world size)
    if distributed:
        dirs = _shard_for_rank(dirs)
    if args.sequential:
        for i, data_dir in enumerate(dirs):
            match(data_dir, i)
    elif dirs:
        match_all(dirs)
    if distributed:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
