"""Registration loop, oracle restatement of reference PointCloud/mlp_reg.py:17-237 (torch CPU).

``train``            mlp_reg.py:17-152  Adam + ReduceLROnPlateau over <=300 epochs of
                     pose repr -> MLP -> pose -> calculate_pc -> L1 Chamfer, best-loss tracking
                     (strict ``<`` on ``loss.item()``, min_loss starts at 1000), early stop when
                     the non-improving count exceeds ``stop`` (checked BEFORE that epoch's update).
``calculate_pc``     mlp_reg.py:155-170
``resample_cluster`` mlp_reg.py:172-237 (non-``normal`` branch)
The module-global ROT of the reference is an explicit ``rot`` argument here; ``epochs`` (300 in
the reference, mlp_reg.py:60) is exposed so parity tests can pin short trajectories.
"""
import numpy as np
import torch

from . import dq as DQ
from . import transforms as T
from .chamfer import chamfer_distance
from .kmeans import k_means


def calculate_pc(local_clusters, matrices):
    return [c @ M[:3, :3].T + M[:3, 3] for c, M in zip(local_clusters, matrices)]


def pose_forward(m, model, rot):
    """One pass pose -> representation -> model -> pose (mlp_reg.py:62-90). Returns (K,4,4)."""
    m2 = m.clone()
    if rot == "dq":
        return DQ.dualquat_to_transform(model(DQ.transform_to_dualquat(m2)))
    R, t = m2[:, :3, :3], m2[:, :3, 3]
    if rot == "q":
        t2, r2 = model(torch.cat([t, T.matrix_to_quaternion(R)], 1))
        R2 = T.quaternion_to_matrix(r2)
    elif rot == "rpy":
        t2, r2 = model(torch.cat([t, T.matrix_to_euler_angles(R, "XYZ")], 1))
        R2 = T.euler_angles_to_matrix(r2, "XYZ")
    elif rot == "6d":
        t2, r2 = model(torch.cat([t, T.matrix_to_rotation_6d(R)], 1))
        R2 = T.rotation_6d_to_matrix(r2)
    else:
        raise ValueError(rot)
    m2[:, :3, :3] = R2
    m2[:, :3, 3] = t2
    return m2


def _train_snapshot(model, opt, sched, epoch, min_loss, count, best_epoch, best_m, best_pcd):
    """The state ENTERING `epoch` (after `epoch` completed epochs), as creg_train_state carries it (include/creg.h): parameters, Adam's
    moments and step count, ReduceLROnPlateau's best / num_bad_epochs / lr, train()'s min_loss / count / best so far."""
    ps = list(model.parameters())
    st = [opt.state[p] for p in ps] if len(opt.state) else None
    return {"params": {n: p.detach().clone() for n, p in model.named_parameters()},
            "exp_avg": {n: (opt.state[p]["exp_avg"].clone() if st else torch.zeros_like(p)) for n, p in model.named_parameters()},
            "exp_avg_sq": {n: (opt.state[p]["exp_avg_sq"].clone() if st else torch.zeros_like(p)) for n, p in model.named_parameters()},
            "step": int(st[0]["step"]) if st else 0, "epochs_run": epoch, "lr": float(opt.param_groups[0]["lr"]),
            "sched_best": float(sched.best), "sched_bad": int(sched.num_bad_epochs), "count": int(count), "min_loss": float(min_loss),
            "best_epoch": int(best_epoch), "stopped": 0,
            "best_m": None if best_m is None else best_m.detach().clone(),
            "best_pred": None if best_pcd is None else torch.cat([p.detach() for p in best_pcd], 0).clone()}


def train(m, y, model, clusters, stop=200, learning_rate=0.0002, scheduler_patience=5,
          scheduler_factor=0.7, rot="q", epochs=300, snapshot_at=(), snapshots=None):
    """snapshot_at / snapshots (test infrastructure of the teacher-forced parity tests): for every epoch e in `snapshot_at` the dict
    `snapshots[e]` receives the state entering epoch e (_train_snapshot) and, once the epoch has run, `loss`, `m2` and the parameter
    gradients of that epoch under "epoch".  Nothing else changes: the trajectory is the plain train()'s."""
    opt = torch.optim.Adam(model.parameters(), lr=learning_rate)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=scheduler_factor,
                                                       patience=scheduler_patience)
    min_loss, best_pcd, best_m, count = 1000, None, None, 0
    best_epoch = -1
    losses, lrs = [], []
    for epoch in range(epochs):
        if epoch in snapshot_at:
            snapshots[epoch] = _train_snapshot(model, opt, sched, epoch, min_loss, count, best_epoch, best_m, best_pcd)
        m2 = pose_forward(m, model, rot)
        pred_list = calculate_pc(clusters, m2)
        loss, _ = chamfer_distance(torch.cat(pred_list, 0).unsqueeze(0), y.unsqueeze(0), norm=1)
        lv = loss.item()
        losses.append(lv)
        lrs.append(opt.param_groups[0]["lr"])
        if lv < min_loss:
            min_loss, best_pcd, best_m, count = lv, pred_list, m2, 0
            best_epoch = epoch
        else:
            count += 1
            if count > stop:
                break
        opt.zero_grad()
        loss.backward()
        if epoch in snapshot_at:
            snapshots[epoch]["epoch"] = {"loss": lv, "m2": m2.detach().clone(),
                                         "grad": {n: p.grad.detach().clone() for n, p in model.named_parameters()}}
        opt.step()
        sched.step(loss)
    if epochs in snapshot_at:
        snapshots[epochs] = _train_snapshot(model, opt, sched, epochs, min_loss, count, best_epoch, best_m, best_pcd)
    pred_np = [p.detach().cpu().numpy() for p in best_pcd]
    return pred_np, best_m, min_loss, {"loss": losses, "lr": lrs}


def resample_cluster(pc_np, n_clusters, matrices):
    """pc_np (N,3) f64 world points of the next frame, matrices (K,4,4): -> (list of K local (M_k,3) f64, labels)."""
    pc_np = np.asarray(pc_np, np.float64)
    matrices = np.asarray(matrices)
    _, labels, _, _ = k_means(pc_np, init=matrices[:, :3, 3], n_clusters=n_clusters, n_init=1)
    out = []
    for i in range(n_clusters):
        pts = pc_np[labels == i]
        inv = np.linalg.inv(matrices[i])
        out.append((inv @ np.hstack([pts, np.ones((len(pts), 1))]).T)[:3].T)
    return out, labels


def match_sequence(frames, step_matrices, step_cluster_np, train_fn, mlp_icp=False, models=("model", "model_rf")):
    """Loop body of ``match`` (mlp_reg.py:265-378) on arrays: frames list of (N,3) f64, frame-0 poses (K,4,4) and
    local clusters.  ``train_fn`` has ``train``'s signature (the real loop above, or the deterministic stub the
    match-level golden was minted with).  Returns ([matrices per frame incl. frame 0], [clusters per frame], losses).

    The details that matter: ``step_cluster_np`` is assigned ONCE (:248/253) -- it is masked_icp's source cloud for
    every frame (:325) while the boxes come from ``pred_pcd_np`` of the re-sampled clusters; "Anchor" always trains
    on the frame-0 clusters (:353); the default branch saves float32 poses, --mlp_icp float64 ones."""
    from .icp import masked_icp
    K = len(step_cluster_np)
    m_t = torch.tensor(np.asarray(step_matrices), dtype=torch.float32)
    cl_t = [torch.tensor(step_cluster_np[i], dtype=torch.float32) for i in range(K)]
    cl_init = [torch.tensor(step_cluster_np[i], dtype=torch.float32) for i in range(K)]
    model, model_rf = models
    mats, clusters, losses = [np.asarray(step_matrices)], [list(step_cluster_np)], []
    for i in range(len(frames) - 1):
        target_np = np.array(frames[i + 1])
        target = torch.tensor(target_np, dtype=torch.float32)
        if mlp_icp:
            pred_np, _, step_m, best = train_fn(m=m_t, y=target, model=model, clusters=cl_t)
            losses.append(best)
            step_m_np = step_m.detach().cpu().numpy()
            _, matrices = masked_icp(step_cluster_np, pred_np, target_np, step_m_np, ori=False)
            new_seg, _ = resample_cluster(target_np, K, matrices)
            cl_t = [torch.tensor(new_seg[j], dtype=torch.float32) for j in range(K)]
            m_t = torch.tensor(matrices, dtype=torch.float32)
            mats.append(matrices)
        else:
            _, _, step_m, _ = train_fn(m=m_t, y=target, model=model, clusters=cl_t)
            m_t = step_m.detach().clone()
            _, _, step_m, best = train_fn(m=m_t, y=target, model=model_rf, clusters=cl_init, learning_rate=0.0001)
            m_t = step_m.detach().clone()
            losses.append(best)
            step_m_np = step_m.detach().cpu().numpy()
            new_seg, _ = resample_cluster(target_np, K, step_m_np)
            cl_t = [torch.tensor(new_seg[j], dtype=torch.float32) for j in range(K)]
            mats.append(step_m_np)
        clusters.append(new_seg)
    return mats, clusters, losses
