"""Device-resident registration of one sequence: the loop body of ``match`` (reference
mlp_reg.py:293-378, default MLP+MLP branch) without the host round trips the reference API
implies (numpy cluster lists, ``loss.item()``), for callers that keep frames in HBM
(bench.py, the multi-GPU driver).  Arithmetic is identical to ``mlp_reg.register_sequence``:
the same train plans (A1) and the same k-means / grouping kernels (K2) run in the same order."""
import torch

from . import mlp_reg, ops
from .model_utils import DQRegMLP, QRegMLP


class SequenceRegistrar:
    def __init__(self, mats0, clusters0, n_tgt, rot="q", hidden=512, epochs=300, use_graph=True, device="cuda",
                 seed=0):
        self.device = torch.device(device)
        self.rot, self.K = rot, len(clusters0)
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(seed)                      # the reference leaves the MLP init unseeded
        ctor = (lambda: QRegMLP(True, hidden_dim=hidden)) if rot == "q" else (lambda: DQRegMLP(hidden_dim=hidden))
        self.model, self.model_rf = ctor().to(self.device), ctor().to(self.device)
        torch.random.set_rng_state(gen_state)
        order = ops.Q_PARAM_ORDER if rot == "q" else ops.DQ_PARAM_ORDER
        self.p_step = [dict(self.model.named_parameters())[n].data for n in order]
        self.p_anchor = [dict(self.model_rf.named_parameters())[n].data for n in order]
        self.m = torch.as_tensor(mats0, dtype=torch.float32).to(self.device).contiguous()
        self.pts, self.off = ops.pack_clusters(clusters0, self.device)
        self.pts_init, self.off_init = self.pts.clone(), self.off.clone()
        self.plan = ops.TrainPlan(rot, self.K, hidden, self.pts.shape[0], n_tgt, epochs=epochs, use_graph=use_graph,
                                  device=self.device)

    def step(self, frame64: torch.Tensor, frame32: torch.Tensor = None):
        """Register the next frame ((N,3) fp64 on the device).  Returns (poses (K,4,4) fp32, result (4))."""
        y = frame32 if frame32 is not None else frame64.to(torch.float32)
        m1, _, _, _, _ = self.plan.run(self.m, y, self.pts, self.off, self.p_step, lr=2e-4)            # "Step"
        m2, _, res, _, _ = self.plan.run(m1, y, self.pts_init, self.off_init, self.p_anchor, lr=1e-4)  # "Anchor"
        M64 = m2.to(torch.float64)
        _, labels, _, _ = ops.kmeans_lloyd(frame64, M64[:, :3, 3].contiguous())
        local, self.off = ops.group_to_local(frame64, labels, M64)
        self.pts = local.to(torch.float32)
        self.m = m2
        return m2, res
