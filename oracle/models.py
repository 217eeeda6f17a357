"""Pose-regression MLPs, oracle restatement of reference PointCloud/model_utils.py:65-168.

Parameter names equal the reference's state_dict keys (encoder.0, decoder_1.{0,2},
decoder_2.{0,2}; decoder.{0,2}) so pinned weights can be exchanged with the reference module.
"""
import torch
from torch import nn


def sincos_features(x: torch.Tensor) -> torch.Tensor:
    """[sin x, cos x, sin 2x, cos 2x, sin 4x, cos 4x, sin 8x, cos 8x] along dim 1 (model_utils.py:141-150)."""
    return torch.cat([f(m * x) for m in (1, 2, 4, 8) for f in (torch.sin, torch.cos)], 1)


class QRegMLP(nn.Module):
    """(K,7)=[t, q_wxyz] -> (t + dt, normalize(q + dq)); model_utils.py:101-159, multi_decoder=True."""

    def __init__(self, multi_decoder: bool = True, hidden_dim: int = 512):
        super().__init__()
        if not multi_decoder:
            raise NotImplementedError("reference single-decoder branch reads an undefined self.add")
        h = hidden_dim
        self.encoder = nn.Sequential(nn.Linear(56, h), nn.LeakyReLU())
        self.decoder_1 = nn.Sequential(nn.Linear(h, h // 2), nn.LeakyReLU(), nn.Linear(h // 2, 3))
        self.decoder_2 = nn.Sequential(nn.Linear(h, h), nn.LeakyReLU(), nn.Linear(h, 4))

    def forward(self, x):
        z = self.encoder(sincos_features(x))
        return self.decoder_1(z) + x[:, :3], nn.functional.normalize(self.decoder_2(z) + x[:, 3:], dim=1)


class DQRegMLP(nn.Module):
    """(K,8) dual quaternion -> residual update, no renormalisation; model_utils.py:65-99."""

    def __init__(self, hidden_dim: int = 512):
        super().__init__()
        h = hidden_dim
        self.decoder = nn.Sequential(nn.Linear(h, h), nn.ReLU(), nn.Linear(h, 8))
        self.encoder = nn.Sequential(nn.Linear(64, h), nn.ReLU())

    def forward(self, x):
        return self.decoder(self.encoder(sincos_features(x))) + x


class RRegMLP(nn.Module):
    """(K,9)=[t, 6d rotation] -> (t + dt, r6d + dr6d); model_utils.py:170-214 (--r 6d)."""

    def __init__(self, hidden_dim: int = 512):
        super().__init__()
        h = hidden_dim
        self.decoder_1 = nn.Sequential(nn.Linear(h, h // 2), nn.LeakyReLU(), nn.Linear(h // 2, 3))
        self.decoder_2 = nn.Sequential(nn.Linear(h, h), nn.LeakyReLU(), nn.Linear(h, 6))
        self.encoder = nn.Sequential(nn.Linear(72, h), nn.LeakyReLU())

    def forward(self, x):
        z = self.encoder(sincos_features(x))
        return self.decoder_1(z) + x[:, :3], self.decoder_2(z) + x[:, 3:]


class RegMLP(nn.Module):
    """(K,6)=[t, rpy] -> (t + dt, rpy + tanh(.)); model_utils.py:216-281 (--r rpy).  The reference
    constructs it as RegMLP(6, 3) (mlp_reg.py:285), i.e. multi_decoder=6 (truthy) and hidden_dim=3."""

    def __init__(self, multi_decoder=True, hidden_dim: int = 512):
        super().__init__()
        if not multi_decoder:
            raise NotImplementedError("single-decoder RegMLP is never constructed on the reference path")
        h = hidden_dim
        self.decoder_1 = nn.Sequential(nn.Linear(h, h // 2), nn.LeakyReLU(), nn.Linear(h // 2, 3))
        self.decoder_2 = nn.Sequential(nn.Linear(h, h), nn.LeakyReLU(), nn.Linear(h, 3), nn.Tanh())
        self.encoder = nn.Sequential(nn.Linear(48, h), nn.LeakyReLU())

    def forward(self, x):
        z = self.encoder(sincos_features(x))
        return self.decoder_1(z) + x[:, :3], self.decoder_2(z) + x[:, 3:]
