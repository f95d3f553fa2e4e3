"""Diffusion pieces on the MLA training path (reference: models/diffusion/): cosine schedule + q_sample
(gaussian_diffusion.py:115-140, 166-229), ActionEmbedder / TimestepEmbedder / LabelEmbedder / FinalLayer
(models.py:28-189; timm 0.9.10 Mlp and RmsNorm restated -- state-dict keys mlp.fc1 / mlp.fc2 / norm_final.weight).
The DiT / ActionModel / sampling loops are dead code for MLA training (SURVEY header) and are not built.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from . import hip, ops
from .llama import Linear


def get_named_beta_schedule(schedule_name: str, num_diffusion_timesteps: int) -> np.ndarray:
    if schedule_name != "squaredcos_cap_v2":
        raise NotImplementedError(f"unknown beta schedule: {schedule_name} (MLA uses squaredcos_cap_v2)")
    alpha_bar = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2  # noqa: E731
    betas = []
    for i in range(num_diffusion_timesteps):
        t1, t2 = i / num_diffusion_timesteps, (i + 1) / num_diffusion_timesteps
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), 0.999))
    return np.array(betas, dtype=np.float64)


class GaussianDiffusion:
    """Training-side subset: float64 tables, q_sample on the GPU."""

    def __init__(self, betas):
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self._dev_tables = {}

    def _tables(self, device):
        key = str(device)
        if key not in self._dev_tables:
            self._dev_tables[key] = (torch.from_numpy(self.sqrt_alphas_cumprod).float().to(device),
                                     torch.from_numpy(self.sqrt_one_minus_alphas_cumprod).float().to(device))
        return self._dev_tables[key]

    def q_sample(self, x_start, t, noise=None):
        """x_t = sqrt(acp[t]) x0 + sqrt(1 - acp[t]) eps; fp32 result like the reference (fp32 table promotes)."""
        if noise is None:
            noise = torch.randn_like(x_start)
        a, b = self._tables(x_start.device)
        return hip.q_sample(x_start.float().contiguous(), noise.float().contiguous(), t.contiguous(), a, b)


def create_diffusion(timestep_respacing="", noise_schedule="squaredcos_cap_v2", diffusion_steps=100, **kwargs):
    """models/diffusion/__init__.py:12-47 for the training configuration MLA uses (no respacing)."""
    if timestep_respacing not in (None, "", [diffusion_steps]):
        raise NotImplementedError("timestep respacing (DDIM sampling) is inference-side (SURVEY 8f rank 2)")
    base = GaussianDiffusion(get_named_beta_schedule(noise_schedule, diffusion_steps))
    # SpacedDiffusion (models/diffusion/respace.py:75-89) re-derives the betas from the base cumulative products even
    # when every step is kept; the training tables come from those re-derived betas.
    last, new_betas = 1.0, []
    for a in base.alphas_cumprod:
        new_betas.append(1 - a / last)
        last = a
    return GaussianDiffusion(np.array(new_betas))


class Mlp(nn.Module):
    """timm.layers.Mlp: fc1 -> act -> (drop) -> (norm = Identity) -> fc2 -> (drop); GELU(tanh) here."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_kind=hip.ACT_GELU_TANH):
        super().__init__()
        self.fc1 = Linear(in_features, hidden_features or in_features)
        self.fc2 = Linear(hidden_features or in_features, out_features or in_features)
        self.act_kind = act_kind

    def forward(self, x):
        return self.fc2(ops.act(self.fc1(x), self.act_kind))


class RmsNorm(nn.Module):
    """timm.layers.RmsNorm(C, eps): x * rsqrt(mean(x^2) + eps) * weight."""

    def __init__(self, channels, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(channels))

    def forward(self, x):
        return ops.rmsnorm(x, self.weight, self.eps)


class ActionEmbedder(nn.Module):  # models.py:112-123
    def __init__(self, action_size, hidden_size):
        super().__init__()
        self.mlp = Mlp(in_features=action_size, hidden_features=hidden_size, out_features=hidden_size)

    def forward(self, x):
        return self.mlp(x)


class TimestepEmbedder(nn.Module):  # models.py:28-65
    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(),
                                 Linear(hidden_size, hidden_size, bias=True))
        self.frequency_embedding_size = frequency_embedding_size

    @staticmethod
    def timestep_embedding(t, dim, max_period=10000):
        half = dim // 2
        freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(device=t.device)
        args = t[:, None].float() * freqs[None]
        return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)

    def forward(self, t):
        t_freq = self.timestep_embedding(t, self.frequency_embedding_size).to(self.mlp[0].weight.dtype)
        return self.mlp[2](ops.act(self.mlp[0](t_freq), hip.ACT_SILU))


class LabelEmbedder(nn.Module):  # models.py:67-97 -- identity when dropout_prob <= 0 (the training setting)
    def __init__(self, in_size, hidden_size, dropout_prob=-1, conditions_shape=(1, 1, 4096)):
        super().__init__()
        self.dropout_prob = dropout_prob
        if dropout_prob > 0:
            raise NotImplementedError("classifier-free-guidance token dropout is not used by any MLA script")

    def forward(self, conditions, train, force_drop_ids=None):
        return conditions


class FinalLayer(nn.Module):  # models.py:173-189
    def __init__(self, hidden_size, out_channels):
        super().__init__()
        self.norm_final = RmsNorm(hidden_size, eps=1e-6)
        self.mlp = Mlp(in_features=hidden_size, hidden_features=hidden_size, out_features=out_channels)

    def forward(self, x):
        return self.mlp(self.norm_final(x))
