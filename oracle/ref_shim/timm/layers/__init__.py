"""Stand-in for the pieces of timm==0.9.10 (pyproject.toml:44) that models/diffusion/models.py:18 imports. timm cannot be
installed in the build image, so each class restates the published v0.9.10 source it names. TEST INFRASTRUCTURE ONLY."""
import torch
import torch.nn as nn


class Mlp(nn.Module):
    """timm.layers.Mlp: fc1 -> act -> drop1 -> norm (Identity) -> fc2 -> drop2."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, norm_layer=None, bias=True,
                 drop=0.0, use_conv=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


def rms_norm(x, normalized_shape, weight=None, eps=1e-5):
    """Restatement of timm tag v0.9.10 `timm/layers/fast_norm.py::rms_norm` (the non-scripting branch; timm is not installable in
    the build image, the reference pins timm==0.9.10 at pyproject.toml:44):

        dims = tuple(range(-1, -norm_ndim - 1, -1));  v = torch.var(x, dim=dims, keepdim=True)
        x = x * torch.rsqrt(v + eps);  if weight is not None: x = x * weight

    `torch.var` defaults to the unbiased (N - 1), MEAN-SUBTRACTED variance, so 0.9.10's "RmsNorm" is not a mean-of-squares
    RMS norm. timm 1.0.13 ("Fix existing RmsNorm layer & fn to match standard formulation ... move old impl to SimpleNorm")
    changed it; the 0.9.10 arithmetic survives there as `simple_norm`. `fast_rms_norm` only takes another route under
    torch.jit scripting (same formula) or with apex installed (`fused_rms_norm_affine`, true RMS) -- the reference does not
    depend on apex (pyproject.toml), so this fallback is what `FinalLayer.norm_final` runs."""
    dims = tuple(range(-1, -len(normalized_shape) - 1, -1))
    v = torch.var(x, dim=dims, keepdim=True)
    x = x * torch.rsqrt(v + eps)
    if weight is not None:
        x = x * weight
    return x


class RmsNorm(nn.Module):
    """timm v0.9.10 `timm/layers/norm.py::RmsNorm(channels, eps=1e-6, affine=True)`: weight initialised to ones;
    forward = fast_rms_norm(x, self.normalized_shape, self.weight, self.eps) -> rms_norm above."""

    def __init__(self, channels, eps=1e-6, affine=True, device=None, dtype=None):
        super().__init__()
        self.normalized_shape = (channels,)
        self.eps = eps
        self.elementwise_affine = affine
        self.weight = nn.Parameter(torch.ones(channels)) if affine else None

    def forward(self, x):
        return rms_norm(x, self.normalized_shape, self.weight, self.eps)


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


class Attention(nn.Module):  # imported by models/diffusion/models.py for the (dead) DiT blocks
    def __init__(self, dim, num_heads=8, qkv_bias=False, **kw):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def lecun_normal_(tensor):
    return nn.init.normal_(tensor, std=(1.0 / tensor.shape[1]) ** 0.5)
