"""Device-resident registration of one sequence: the loop body of ``match`` (reference
mlp_reg.py:293-378, default MLP+MLP branch) without the host round trips the reference API
implies (numpy cluster lists, ``loss.item()``), for callers that keep frames in HBM
(bench.py, the multi-GPU driver).  Arithmetic is identical to ``mlp_reg.register_sequence``:
the same train plans (A1) and the same k-means / grouping kernels (K2) run in the same order."""
import numpy as np
import torch

from . import _lib, mlp_reg, ops
from .model_utils import DQRegMLP, QRegMLP, RegMLP, RRegMLP


class _HostInverse:
    """inv(M_k) the way resample_cluster gets it (mlp_reg.py:211): ``np.linalg.inv`` on the HOST in the poses' own
    dtype (float32 after train, float64 after masked_icp) -- LAPACK's rounding is what the reference's
    cluster/NNNN.npz carries and it is not reproducible on the device.  The (K,4,4) poses are fetched on a side
    stream once the event recorded right after the launch that produced them has completed, so work enqueued in
    between (the k-means of the same frames) overlaps the round trip; returns the inverses as float64 on the device.

    The HOST waits for that event (hipEventSynchronize), not the side stream: `side.wait_event(ev)` parks a barrier packet in
    the side stream's hardware queue for as long as the trains run, and the command processor's polling of it slows the
    latency-bound epoch chains in the other queue by 3-10 % (round 4, tests/measure/frame_phases_by_chains.py: a train of
    5 problems 14.27 -> 13.82 ms with two graph chains, 15.64 -> 14.07 ms as one linear graph)."""

    def __init__(self, device):
        self.side = torch.cuda.Stream(device=device)

    def mark(self):
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def __call__(self, M, ev):
        ev.synchronize()
        with torch.cuda.stream(self.side):
            host = M.to("cpu", non_blocking=True)
            self.side.synchronize()
        inv = np.linalg.inv(host.numpy()).astype(np.float64)
        return torch.from_numpy(inv).to(M.device, non_blocking=True)


class SequenceRegistrar:
    def __init__(self, mats0, clusters0, n_tgt, rot="q", hidden=512, epochs=300, use_graph=True, device=None,
                 seed=0):
        self._init_state(mats0, clusters0, rot, hidden, _lib.device(torch.device(device) if device is not None else None), seed)
        self.plan = ops.TrainPlan(rot, self.K, hidden, self.pts.shape[0], n_tgt, epochs=epochs, use_graph=use_graph,
                                  device=self.device)

    def _init_state(self, mats0, clusters0, rot, hidden, device, seed, models=None):
        self.device = device
        self.rot, self.K = rot, len(clusters0)
        if models is not None:                       # caller-made (model, model_rf), e.g. the drop-in's _make_models()
            self.model, self.model_rf = models[0].to(self.device), models[1].to(self.device)
        else:
            gen_state = torch.random.get_rng_state()
            torch.manual_seed(seed)                  # the reference leaves the MLP init unseeded
            ctor = {"q": lambda: QRegMLP(True, hidden_dim=hidden), "dq": lambda: DQRegMLP(hidden_dim=hidden),
                    "6d": lambda: RRegMLP(hidden_dim=hidden), "rpy": lambda: RegMLP(True, hidden_dim=hidden)}[rot]
            self.model, self.model_rf = ctor().to(self.device), ctor().to(self.device)
            torch.random.set_rng_state(gen_state)
        order = ops.DQ_PARAM_ORDER if rot == "dq" else ops.Q_PARAM_ORDER
        self.p_step = [dict(self.model.named_parameters())[n].data for n in order]
        self.p_anchor = [dict(self.model_rf.named_parameters())[n].data for n in order]
        self.m = torch.as_tensor(mats0, dtype=torch.float32).to(self.device).contiguous()
        self.pts, self.off = ops.pack_clusters(clusters0, self.device)
        self.pts_init, self.off_init = self.pts.clone(), self.off.clone()
        self.local64, _ = ops.pack_clusters(clusters0, self.device, torch.float64)   # what resample_cluster returned last
        self.local64_init = self.local64             # step_cluster_np of the reference: assigned once (mlp_reg.py:248/253)
        self.host_inverse = _HostInverse(self.device)

    def step(self, frame64: torch.Tensor, frame32: torch.Tensor = None):
        """Register the next frame ((N,3) fp64 on the device).  Returns (poses (K,4,4) fp32, result (4))."""
        y = frame32 if frame32 is not None else frame64.to(torch.float32)
        m1, _, _, _, _ = self.plan.run(self.m, y, self.pts, self.off, self.p_step, lr=2e-4)            # "Step"
        m2, _, res, _, _ = self.plan.run(m1, y, self.pts_init, self.off_init, self.p_anchor, lr=1e-4, same_target=True)  # "Anchor": the frame's k-d leaves are Step's
        ev = self.host_inverse.mark()
        _, labels, _, _ = ops.kmeans_lloyd(frame64, m2[:, :3, 3].to(torch.float64).contiguous())
        local, self.off = ops.group_to_local(frame64, labels, self.host_inverse(m2, ev), m_is_inverse=True)
        self.pts = local.to(torch.float32)
        self.m = m2
        return m2, res


class BatchRegistrar:
    """S independent sequences of identical shape registered in lock-step: frame t of every sequence
    is one batched train plan run (Step), one more (Anchor), then the S re-segmentations.  Sequences
    are independent in the reference (only frames inside a sequence depend on each other), so the
    results are those of S separate SequenceRegistrars -- bit for bit -- while every launch carries S
    problems and the latency-bound kernels of one sequence hide behind the others'."""

    def __init__(self, mats0, clusters0, n_tgt, n_sequences, rot="q", hidden=512, epochs=300, use_graph=True,
                 device=None, seeds=None, models=None, graph_branches=0, nn_search=0):
        self.device = _lib.device(torch.device(device) if device is not None else None)
        self.S = n_sequences
        self.host_inverse = _HostInverse(self.device)
        seeds = list(seeds) if seeds is not None else list(range(n_sequences))
        self.seqs = []
        for s in range(n_sequences):
            r = SequenceRegistrar.__new__(SequenceRegistrar)
            SequenceRegistrar._init_state(r, mats0, clusters0, rot, hidden, self.device, seeds[s],
                                          None if models is None else models[s])
            self.seqs.append(r)
        self.plan = ops.TrainPlan(rot, len(clusters0), hidden, self.seqs[0].pts.shape[0], n_tgt, epochs=epochs,
                                  use_graph=use_graph, device=self.device, batch=n_sequences, graph_branches=graph_branches,
                                 nn_search=nn_search)

    def _train(self, problems, lr, same_target=False):
        """One batched `train` (mlp_reg.py:17-152) of the S problems (m, y, pts, offsets, params); a seam so the
        match-level golden test can replay the reference's loop with the deterministic stub the golden was minted
        with (tests/_match_stub.py).  Returns per problem (best_m, best_pred, result, ...).  same_target: "Anchor" right
        after "Step" on the same frames -- the plan keeps the frames' k-d leaf blocks."""
        return self.plan.run_batch(problems, lr=lr, stop=getattr(self, "stop", 200), same_target=same_target)

    normal = False      # --normal (mlp_reg.py:190-203): the re-segmentation clusters [xyz | 0.5 n]

    def _kmeans(self, frames64, inits):
        """The S re-segmentations of a round (resample_cluster's k_means, mlp_reg.py:187-204): one launch for all sequences where the
        frames fit, per sequence otherwise; under `normal` the 6-D form per sequence (normals: GPU neighbour search + host orientation)."""
        if self.normal:
            from .normals import point_features
            km = []
            for f, c in zip(frames64, inits):
                feat, _ = point_features(f.cpu().numpy())
                X6 = torch.as_tensor(feat, dtype=torch.float64, device=f.device).contiguous()
                km.append(ops.kmeans_lloyd_nd(X6, torch.cat([c, torch.zeros_like(c)], 1).contiguous()))
            return km
        if frames64[0].shape[0] <= ops.KMEANS_BATCH_MAX_N and self.S <= 16 and inits[0].shape[0] <= 128:      # all S in one launch
            return ops.kmeans_lloyd_batch(frames64, inits)
        return [ops.kmeans_lloyd(f, c) for f, c in zip(frames64, inits)]

    def step(self, frames64, frames32=None):
        """frames64: list of S (N,3) fp64 device tensors (the next frame of every sequence)."""
        ys = frames32 if frames32 is not None else [f.to(torch.float32) for f in frames64]
        step = self._train([(r.m, y, r.pts, r.off, r.p_step) for r, y in zip(self.seqs, ys)], lr=2e-4)
        anchor = self._train([(o[0], y, r.pts_init, r.off_init, r.p_anchor)
                              for r, y, o in zip(self.seqs, ys, step)], lr=1e-4, same_target=True)
        out = []
        # epochs each train ran (result[1]; < the plan's epochs after an early stop) -- bench.py reports them
        self.last_epochs = (torch.stack([o[2][1] for o in step]), torch.stack([o[2][1] for o in anchor]))
        M_all = torch.stack([o[0] for o in anchor])                               # (S,K,4,4) float32: what train returned
        ev = self.host_inverse.mark()
        t_all = M_all[:, :, :3, 3].to(torch.float64).contiguous()                 # one cast and one slice for all sequences
        inits = [t_all[i] for i in range(self.S)]
        km = self._kmeans(frames64, inits)
        inv_all = self.host_inverse(M_all, ev)                                    # float32 LAPACK inverse, as mlp_reg.py:211
        invs = [inv_all[i] for i in range(self.S)]
        if self.S <= ops.GROUP_BATCH_MAX and len({f.shape[0] for f in frames64}) == 1:
            groups = ops.group_to_local_batch(frames64, [res[1] for res in km], invs, m_is_inverse=True)     # all S in one launch pair
        else:
            groups = [ops.group_to_local(f, res[1], I, m_is_inverse=True) for f, res, I in zip(frames64, km, invs)]
        for r, o, (local, off) in zip(self.seqs, anchor, groups):
            m2 = o[0]
            r.off = off
            r.local64 = local                        # what resample_cluster returns (the cluster/NNNN.npz contents)
            r.pts, r.m = local.to(torch.float32), m2
            out.append((m2, o[2]))
        return out


def _step_mlp_icp(self, frames64, frames32=None):
    """The `--mlp_icp` frame of match() (mlp_reg.py:296-332) for all S sequences: ONE batched "Step" train on the
    current (re-sampled) clusters, ONE masked-ICP launch (clusters x sequences) started from the trained poses whose
    SOURCES are the frame-0 clusters (the reference never reassigns `step_cluster_np`, mlp_reg.py:248,325) and whose
    mask boxes are those of the trained clouds of the current segmentation (`pred_pcd_np`), ONE k-means launch.
    Returns [(poses (K,4,4) fp64, train result (4))] per sequence."""
    ys = frames32 if frames32 is not None else [f.to(torch.float32) for f in frames64]
    step = self._train([(r.m, y, r.pts, r.off, r.p_step) for r, y in zip(self.seqs, ys)], lr=2e-4)
    probs = [(r.local64_init, o[1], r.off_init, f, o[0].to(torch.float64), r.off) for r, f, o in zip(self.seqs, frames64, step)]
    if len(probs) <= ops.ICP_BATCH_MAX:
        icp = ops.masked_icp_batch(probs)
    else:
        icp = [ops.masked_icp(*p) for p in probs]
    Ms = [M for M, _, _ in icp]
    M_all = torch.stack(Ms)                                                       # (S,K,4,4) float64: masked_icp's poses
    ev = self.host_inverse.mark()
    inits = [M[:, :3, 3].contiguous() for M in Ms]
    km = self._kmeans(frames64, inits)
    out = []
    inv_all = self.host_inverse(M_all, ev)                                        # float64 LAPACK inverse (mlp_reg.py:211,326)
    invs = [inv_all[i] for i in range(self.S)]
    if len(self.seqs) <= ops.GROUP_BATCH_MAX and len({f.shape[0] for f in frames64}) == 1:
        groups = ops.group_to_local_batch(frames64, [res[1] for res in km], invs, m_is_inverse=True)
    else:
        groups = [ops.group_to_local(f, res[1], I, m_is_inverse=True) for f, res, I in zip(frames64, km, invs)]
    for r, M, o, (local, off) in zip(self.seqs, Ms, step, groups):
        r.local64, r.off = local, off
        r.pts, r.m = r.local64.to(torch.float32), M.to(torch.float32)
        out.append((M, o[2]))
    return out


BatchRegistrar.step_mlp_icp = _step_mlp_icp


class IcpRegistrar:
    """The ICP-style frame of the north star (SURVEY 8(d) second line), device resident, no MLP:
    K3 cluster_transform (current world clusters = mask boxes) -> K4 masked point-to-point ICP of every
    cluster against the new frame (cluster_icp.py:118-191) -> K5 poses as dual quaternions
    (dq_func.py:100-124) -> K2 Lloyd k-means seeded at the new translations + change of frame
    (resample_cluster, mlp_reg.py:172-237).  The same kernels the `--mlp_icp` branch of match() runs
    (mlp_reg.py:325-326), minus train."""

    def __init__(self, mats0, clusters0, device=None):
        self.device = _lib.device(torch.device(device) if device is not None else None)
        self.M = torch.as_tensor(mats0, dtype=torch.float64).to(self.device).contiguous()
        self.local, self.off = ops.pack_clusters(clusters0, self.device, torch.float64)

    def step(self, frame64: torch.Tensor):
        """Returns (poses (K,4,4) fp64, dual quaternions (K,8) fp32, ICP iterations (K) int32)."""
        world32 = ops.cluster_transform(self.local.to(torch.float32), self.off, self.M.to(torch.float32))
        M_new, _, n_it = ops.masked_icp(self.local, world32, self.off, frame64, self.M)
        dq = ops.se3_to_dq(M_new.to(torch.float32))
        _, labels, _, _ = ops.kmeans_lloyd(frame64, M_new[:, :3, 3].contiguous())
        self.local, self.off = ops.group_to_local(frame64, labels, M_new)
        self.M = M_new
        return M_new, dq, n_it


class BatchIcpRegistrar:
    """S sequences of identical frame size through IcpRegistrar's steps in lock-step: the S ICP launches
    share one launch (grid clusters x sequences) and the S re-segmentations one k-means launch (no host sync in a
    round).  Results equal S separate IcpRegistrars (the batched k-means is bit-identical to the
    multi-launch one)."""

    def __init__(self, mats0, clusters0, n_sequences, device=None):
        self.regs = [IcpRegistrar(mats0, clusters0, device) for _ in range(n_sequences)]

    def step(self, frames64):
        if len(self.regs) <= ops.ICP_BATCH_MAX and len({(r.local.shape[0], f.shape[0]) for r, f in zip(self.regs, frames64)}) == 1:
            # all sequences' clusters in one launch; the mask boxes (K3 of the current poses) are evaluated inside it
            icp = ops.masked_icp_batch([(r.local, None, r.off, f, r.M) for r, f in zip(self.regs, frames64)])
        else:
            icp = [ops.masked_icp(r.local, ops.cluster_transform(r.local.to(torch.float32), r.off, r.M.to(torch.float32)),
                                  r.off, f, r.M) for r, f in zip(self.regs, frames64)]
        k = icp[0][0].shape[0]
        M_all = torch.stack([M_new for M_new, _, _ in icp])                       # (S,K,4,4): one cast, one K5 launch, one slice
        dq_all = ops.se3_to_dq(M_all.to(torch.float32).reshape(-1, 4, 4)).reshape(len(icp), k, 8)
        t_all = M_all[:, :, :3, 3].contiguous()
        res = [(M_new, dq_all[i], n_it) for i, (M_new, _, n_it) in enumerate(icp)]
        inits = [t_all[i] for i in range(len(icp))]
        if frames64[0].shape[0] <= ops.KMEANS_BATCH_MAX_N and len(self.regs) <= 16 and k <= 128:
            km = ops.kmeans_lloyd_batch(frames64, inits)
        else:
            km = [ops.kmeans_lloyd(f, c) for f, c in zip(frames64, inits)]
        if len(self.regs) <= ops.GROUP_BATCH_MAX and len({f.shape[0] for f in frames64}) == 1:
            groups = ops.group_to_local_batch(frames64, [kr[1] for kr in km], [o[0] for o in res])
        else:
            groups = [ops.group_to_local(f, kr[1], o[0]) for f, o, kr in zip(frames64, res, km)]
        for r, o, (local, off) in zip(self.regs, res, groups):
            r.local, r.off = local, off
            r.M = o[0]
        return res

