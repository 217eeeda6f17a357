"""Teacher-forced late-epoch parity: the oracle -- bit for bit on the reference's float32 trajectory -- runs `train`
(mlp_reg.py:17-152) on the host and hands its state ENTERING epoch e to the HIP plan (creg_train_plan_resume), which runs ONE epoch;
what that epoch produces is compared with the oracle's own epoch e.  Errors cannot accumulate, so the Adam bias corrections at
t ~ 150-300, the plateau scheduler after several cuts and the best tracking late in a train are checked as tightly as epoch 1 is.
Shared by tests/test_gpu_parity.py and tests/measure/teacher_forced.py (which prints the statistics the tolerances were taken from)."""
import numpy as np
import torch


def split(flat, offsets):
    return [flat[offsets[i]:offsets[i + 1]] for i in range(len(offsets) - 1)]


def oracle_snapshots(g, sd, k, epochs, threads=16, rot="q", hidden=512, stop=200, lr=2e-4):
    """The oracle's train on the golden problem `g` (m, y, local, offsets) from the pinned parameters `sd`: snapshots entering every
    epoch of `epochs` and of the epoch after each (the expected state)."""
    from oracle import _clib, models, registration
    want = sorted({e for e in epochs} | {e + 1 for e in epochs})
    keep = torch.get_num_threads()
    torch.set_num_threads(threads)
    _clib.lib().oracle_set_threads(threads)
    try:
        model = models.QRegMLP(True, hidden) if rot == "q" else models.DQRegMLP(hidden)
        model.load_state_dict({kk: v.clone() for kk, v in sd.items()})
        m = torch.from_numpy(np.asarray(g["m"], np.float32))
        y = torch.from_numpy(np.asarray(g["y"], np.float32))
        cl = [torch.from_numpy(np.asarray(c, np.float32)) for c in split(g["local"], g["offsets"])]
        assert len(cl) == k
        snaps = {}
        _, _, _, hist = registration.train(m, y, model, cl, stop=stop, learning_rate=lr, rot=rot, epochs=max(want), snapshot_at=set(want), snapshots=snaps)
    finally:
        torch.set_num_threads(keep)
        _clib.lib().oracle_set_threads(keep)
    return snaps, hist


def plan_epoch(plan, probe_plan, dev, order, m, y, pts, off, s0, stop=200):
    """ONE epoch of the HIP plan from the oracle's state s0.  Returns (pose of the epoch, loss of the epoch, parameters after, state after)."""
    params = [s0["params"][kk].clone().to(dev) for kk in order]
    m2, _, _, _ = probe_plan.probe(m, y, pts, off, params)          # the forward of the entering parameters (before the step)
    state = {kk: s0[kk] for kk in ("step", "epochs_run", "lr", "sched_best", "sched_bad", "count", "min_loss", "best_epoch", "stopped")}
    state["exp_avg"] = [s0["exp_avg"][kk].clone().to(dev) for kk in order]
    state["exp_avg_sq"] = [s0["exp_avg_sq"][kk].clone().to(dev) for kk in order]
    best = None if s0["best_m"] is None else (s0["best_m"].to(dev), s0["best_pred"].to(dev))
    bm, bp, res, lh, lrh, after = plan.resume(m, y, pts, off, params, state, 1, stop=stop, best=best)
    return m2, float(lh[s0["epochs_run"]]), params, after, (bm, bp, res, lrh)


def compare(order, s0, s1, m2, loss, params, after, extra):
    """Statistics of one teacher-forced epoch against the oracle's."""
    bm, bp, res, lrh = extra
    ep = s0["epoch"]
    out = {"loss_rel": abs(loss - ep["loss"]) / abs(ep["loss"]),
           "pose": float((m2.cpu() - ep["m2"]).abs()[:, :3, :].max()),
           "lr_used_exact": float(lrh[s0["epochs_run"]]) == float(np.float32(s0["lr"])),
           "exact": {kk: (after[kk], s1[kk]) for kk in ("step", "epochs_run", "sched_bad", "count", "best_epoch", "stopped")},
           "lr": (after["lr"], s1["lr"]), "sched_best": (after["sched_best"], s1["sched_best"]), "min_loss": (after["min_loss"], s1["min_loss"])}
    upd, mom, msq, masked, total = 0.0, 0.0, 0.0, 0, 0
    worst = None
    for i, kk in enumerate(order):
        g = ep["grad"][kk].reshape(-1)
        live = g.abs() >= 1e-8
        d_ref = (s1["params"][kk] - s0["params"][kk]).reshape(-1)
        d_hip = (params[i].cpu() - s0["params"][kk]).reshape(-1)
        err = (d_hip - d_ref).abs()
        total += g.numel()
        masked += int((~live).sum())
        if live.any():
            e = float(err[live].max())
            if e > upd:
                upd, worst = e, kk
        # the moments themselves (teacher-forced: 0.1 / 0.001 of this epoch's gradient error), relative to the tensor's largest
        ea = (after["exp_avg"][i].cpu() - s1["exp_avg"][kk]).abs().max() / max(float(s1["exp_avg"][kk].abs().max()), 1e-30)
        es = (after["exp_avg_sq"][i].cpu() - s1["exp_avg_sq"][kk]).abs().max() / max(float(s1["exp_avg_sq"][kk].abs().max()), 1e-30)
        mom, msq = max(mom, float(ea)), max(msq, float(es))
        out.setdefault("upd_all", 0.0)
        out["upd_all"] = max(out["upd_all"], float(err.max()))
    out.update({"upd": upd, "upd_worst_tensor": worst, "exp_avg_rel": mom, "exp_avg_sq_rel": msq, "masked_frac": masked / total})
    if s1["best_epoch"] >= 0:
        out["best_m"] = float((bm.cpu() - s1["best_m"]).abs()[:, :3, :].max())
        out["best_pred"] = float((bp.cpu() - s1["best_pred"]).abs().max())
    return out
