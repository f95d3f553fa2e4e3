"""Diffusion pieces on the MLA training path (reference: models/diffusion/): cosine schedule + q_sample
(gaussian_diffusion.py:115-140, 166-229), ActionEmbedder / TimestepEmbedder / LabelEmbedder / FinalLayer
(models.py:28-189; timm 0.9.10 Mlp and RmsNorm restated -- state-dict keys mlp.fc1 / mlp.fc2 / norm_final.weight).
The DDIM / DDPM sampling loops serve inference (model_mla.py:592-775); DiT / ActionModel are dead code for MLA and not built.
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


def space_timesteps(num_timesteps: int, section_counts) -> set:
    """models/diffusion/respace.py:12-66. "ddimN": the first integer stride that yields exactly N steps (N=1 -> {50})."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            if want == 1:
                return {50}
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += frac
        start += size
    return set(steps)


class GaussianDiffusion:
    """Epsilon-prediction, fixed-small-variance Gaussian diffusion (models/diffusion/gaussian_diffusion.py, the configuration
    create_diffusion builds for MLA: learn_sigma=False, sigma_small=True, predict_xstart=False). float64 tables; q_sample runs
    a HIP kernel; the sampling loops update [B, T, action_dim] tensors (a few dozen elements) with torch ops between model calls.
    ``timestep_map`` maps this process's step index to the base process's timestep (respace.py:75-129)."""

    def __init__(self, betas, timestep_map=None, original_num_steps=None):
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        self.timestep_map = list(range(self.num_timesteps)) if timestep_map is None else list(timestep_map)
        self.original_num_steps = self.num_timesteps if original_num_steps is None else original_num_steps
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = (np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
                                               if self.num_timesteps > 1 else np.array([]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
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

    # ------------------------------------------------------------------ sampling (inference side)
    @staticmethod
    def _at(table: np.ndarray, t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
        """_extract_into_tensor gaussian_diffusion.py:862-875: float64 table -> float32 value per batch element, broadcast."""
        v = torch.from_numpy(table).to(device=t.device)[t].float()
        return v.view(-1, *([1] * (like.dim() - 1))).expand(like.shape)

    def _eps(self, model, x, t, model_kwargs):
        """_WrappedModel.__call__ respace.py:118-129 (maps the step index to the base timestep) + the tuple convention of
        p_mean_variance gaussian_diffusion.py:279-283 (PrismaticVLM.forward returns (output, noise_pred) in eval mode)."""
        ts = torch.tensor(self.timestep_map, device=t.device, dtype=t.dtype)[t]
        out = model(x, ts, **(model_kwargs or {}))
        if isinstance(out, tuple):
            out = out[1]
        return out.float()        # bf16 -> fp32 is exact; the fp32 tables promote the arithmetic in the reference as well

    def _pred_xstart(self, x, t, eps, clip_denoised):
        px = self._at(self.sqrt_recip_alphas_cumprod, t, x) * x - self._at(self.sqrt_recipm1_alphas_cumprod, t, x) * eps
        return px.clamp(-1, 1) if clip_denoised else px

    def ddim_sample(self, model, x, t, clip_denoised=True, model_kwargs=None, eta=0.0):
        """gaussian_diffusion.py:520-568 (Song et al. eq. 12)."""
        pred_xstart = self._pred_xstart(x, t, self._eps(model, x, t, model_kwargs), clip_denoised)
        eps = (self._at(self.sqrt_recip_alphas_cumprod, t, x) * x - pred_xstart) / self._at(self.sqrt_recipm1_alphas_cumprod, t, x)
        ab, ab_prev = self._at(self.alphas_cumprod, t, x), self._at(self.alphas_cumprod_prev, t, x)
        sigma = eta * torch.sqrt((1 - ab_prev) / (1 - ab)) * torch.sqrt(1 - ab / ab_prev)
        noise = torch.randn_like(x)              # drawn even when eta == 0, like the reference (keeps the RNG stream aligned)
        mean_pred = pred_xstart * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev - sigma ** 2) * eps
        nonzero = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
        return {"sample": mean_pred + nonzero * sigma * noise, "pred_xstart": pred_xstart}

    def p_sample(self, model, x, t, clip_denoised=True, model_kwargs=None):
        """gaussian_diffusion.py:395-440 with the fixed-small posterior variance."""
        pred_xstart = self._pred_xstart(x, t, self._eps(model, x, t, model_kwargs), clip_denoised)
        mean = self._at(self.posterior_mean_coef1, t, x) * pred_xstart + self._at(self.posterior_mean_coef2, t, x) * x
        logvar = self._at(self.posterior_log_variance_clipped, t, x)
        noise = torch.randn_like(x)
        nonzero = (t != 0).float().view(-1, *([1] * (x.dim() - 1)))
        return {"sample": mean + nonzero * torch.exp(0.5 * logvar) * noise, "pred_xstart": pred_xstart}

    def _loop(self, step, model, shape, noise, device, **kw):
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else torch.randn(*shape, device=device)
        with torch.no_grad():
            for i in reversed(range(self.num_timesteps)):
                t = torch.tensor([i] * shape[0], device=img.device)
                img = step(model, img, t, **kw)["sample"]
        return img

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                         device=None, progress=False, eta=0.0):
        """gaussian_diffusion.py:608-688."""
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn are never passed by MLA (model_mla.py:742-763)")
        return self._loop(self.ddim_sample, model, shape, noise, device, clip_denoised=clip_denoised, model_kwargs=model_kwargs, eta=eta)

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                      device=None, progress=False):
        """gaussian_diffusion.py:442-518."""
        if denoised_fn is not None or cond_fn is not None:
            raise NotImplementedError("denoised_fn / cond_fn are never passed by MLA (model_mla.py:742-763)")
        return self._loop(self.p_sample, model, shape, noise, device, clip_denoised=clip_denoised, model_kwargs=model_kwargs)


def create_diffusion(timestep_respacing="", noise_schedule="squaredcos_cap_v2", diffusion_steps=100, **kwargs):
    """models/diffusion/__init__.py:12-47: SpacedDiffusion over the cosine schedule. Training uses no respacing; inference uses
    "ddimN" (model_mla.py:1166-1173)."""
    base = GaussianDiffusion(get_named_beta_schedule(noise_schedule, diffusion_steps))
    if timestep_respacing is None or timestep_respacing == "":
        timestep_respacing = [diffusion_steps]
    use = space_timesteps(diffusion_steps, timestep_respacing)
    # SpacedDiffusion (respace.py:75-89) re-derives the betas from the base cumulative products of the kept steps -- even when
    # every step is kept; all tables come from those re-derived betas.
    last, new_betas, tmap = 1.0, [], []
    for i, a in enumerate(base.alphas_cumprod):
        if i in use:
            new_betas.append(1 - a / last)
            last = a
            tmap.append(i)
    return GaussianDiffusion(np.array(new_betas), timestep_map=tmap, original_num_steps=diffusion_steps)


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
    """timm==0.9.10 `RmsNorm(channels, eps)` (pin: pyproject.toml:44; imported at models/diffusion/models.py:18).

    timm tag v0.9.10: `timm/layers/norm.py::RmsNorm.forward` calls `fast_rms_norm(x, normalized_shape, weight, eps)`
    (`timm/layers/fast_norm.py`), which without apex falls through to `rms_norm`:
        v = torch.var(x, dim=dims, keepdim=True);  x = x * torch.rsqrt(v + eps);  x = x * weight
    `torch.var` is the UNBIASED, mean-subtracted variance -- this is not the mean-of-squares RMS norm (timm 1.0.13 fixed
    `RmsNorm` and kept this arithmetic under the name `SimpleNorm`). With apex installed timm 0.9.10 would dispatch to
    `fused_rms_norm_affine` (true RMS); the reference's environment (pyproject.toml) does not install apex, so the
    torch.var form is the one matched here."""

    def __init__(self, channels, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(channels))

    def forward(self, x):
        return ops.timm_rmsnorm(x, self.weight, self.eps)


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

    _freqs = {}      # (half, max_period, device) -> the frequency table on that device (same values: computed on the host in fp32 once)

    @staticmethod
    def timestep_embedding(t, dim, max_period=10000):
        half = dim // 2
        key = (half, max_period, str(t.device))
        freqs = TimestepEmbedder._freqs.get(key)
        if freqs is None:
            freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(device=t.device)
            TimestepEmbedder._freqs[key] = freqs
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
