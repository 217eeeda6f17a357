"""Pose-regression MLPs (drop-in for reference PointCloud/model_utils.py:65-168).

Same class names, constructor arguments and state_dict keys as the reference, so checkpoints and
``match()`` are interchangeable.  ``train`` never calls ``forward`` on the hot path: it hands the
parameter tensors to the fused HIP train plan (libcreg.so), which implements exactly this
forward, its backward and Adam.  ``forward`` is kept for API compatibility (inspection, export).
"""
import torch
from torch import nn


def _sincos(x):
    return torch.cat([f(s * x) for s in (1, 2, 4, 8) for f in (torch.sin, torch.cos)], dim=1)


class QRegMLP(nn.Module):
    """[t | q_wxyz] (K,7) -> (t + dt (K,3), normalize(q + dq) (K,4)); model_utils.py:101-159."""

    def __init__(self, multi_decoder=True, hidden_dim=512):
        super().__init__()
        if not multi_decoder:
            raise NotImplementedError("the reference's single-decoder QRegMLP branch reads an undefined "
                                      "attribute (model_utils.py:163); only multi_decoder=True is usable")
        self.multi_decoder, self.hidden_dim = True, hidden_dim
        self.input_dim, self.output_dim_1, self.output_dim_2, self.freq = 7, 3, 4, 4
        h = hidden_dim
        self.decoder_1 = nn.Sequential(nn.Linear(h, h // 2), nn.LeakyReLU(), nn.Linear(h // 2, 3))
        self.decoder_2 = nn.Sequential(nn.Linear(h, h), nn.LeakyReLU(), nn.Linear(h, 4))
        self.encoder = nn.Sequential(nn.Linear(self.input_dim * 8, h), nn.LeakyReLU())

    sin_encoding = staticmethod(_sincos)

    def forward(self, x):
        z = self.encoder(_sincos(x))
        return self.decoder_1(z) + x[:, :3], nn.functional.normalize(self.decoder_2(z) + x[:, 3:], dim=1)


class DQRegMLP(nn.Module):
    """dual quaternion (K,8) -> residual update (K,8), not renormalised; model_utils.py:65-99."""

    def __init__(self, hidden_dim=512):
        super().__init__()
        self.input_dim, self.output_dim, self.hidden_dim, self.freq = 8, 8, hidden_dim, 4
        h = hidden_dim
        self.decoder = nn.Sequential(nn.Linear(h, h), nn.ReLU(), nn.Linear(h, 8))
        self.encoder = nn.Sequential(nn.Linear(self.input_dim * 8, h), nn.ReLU())

    sin_encoding = staticmethod(_sincos)

    def forward(self, x):
        return self.decoder(self.encoder(_sincos(x))) + x


class RRegMLP(nn.Module):
    """[t | 6-D rotation] (K,9) -> (t + dt, r6d + dr); model_utils.py:170-214 (--r 6d).  Trained by the same
    device-resident plan as the other three (rot code 2: `k_head<., true>` takes the ninth output unit, k_bd's X instance the 72 input features)."""

    def __init__(self, hidden_dim=512):
        super().__init__()
        self.add, self.input_dim, self.output_dim_1, self.output_dim_2, self.hidden_dim = True, 9, 3, 6, hidden_dim
        h = hidden_dim
        self.decoder_1 = nn.Sequential(nn.Linear(h, h // 2), nn.LeakyReLU(), nn.Linear(h // 2, 3))
        self.decoder_2 = nn.Sequential(nn.Linear(h, h), nn.LeakyReLU(), nn.Linear(h, 6))
        self.encoder = nn.Sequential(nn.Linear(self.input_dim * 8, h), nn.LeakyReLU())

    sin_encoding = staticmethod(_sincos)

    def forward(self, x):
        z = self.encoder(_sincos(x))
        return self.decoder_1(z) + x[:, :3], self.decoder_2(z) + x[:, 3:]


class RegMLP(nn.Module):
    """[t | rpy] (K,6) -> (t + dt, rpy + tanh(.)); model_utils.py:216-281 (--r rpy).  NB the reference
    builds it as RegMLP(6, 3) (mlp_reg.py:285): multi_decoder=6 is merely truthy and hidden_dim is 3.
    That call is reproduced as is (a latent bug that must not be fixed silently)."""

    def __init__(self, multi_decoder=True, hidden_dim=512):
        super().__init__()
        if not multi_decoder:
            raise NotImplementedError("single-decoder RegMLP is never constructed on the reference path")
        self.multi_decoder, self.input_dim, self.hidden_dim, self.freq = multi_decoder, 6, hidden_dim, 4
        h = hidden_dim
        self.decoder_1 = nn.Sequential(nn.Linear(h, h // 2), nn.LeakyReLU(), nn.Linear(h // 2, 3))
        self.decoder_2 = nn.Sequential(nn.Linear(h, h), nn.LeakyReLU(), nn.Linear(h, 3), nn.Tanh())
        self.encoder = nn.Sequential(nn.Linear(self.input_dim * 8, h), nn.LeakyReLU())

    sin_encoding = staticmethod(_sincos)

    def forward(self, x):
        z = self.encoder(_sincos(x))
        return self.decoder_1(z) + x[:, :3], self.decoder_2(z) + x[:, 3:]
