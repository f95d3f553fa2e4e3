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


class RmsNorm(nn.Module):
    """timm.layers.RmsNorm: x * rsqrt(mean(x^2, -1) + eps) * weight (fast_rms_norm fallback path, computed in x's dtype)."""

    def __init__(self, channels, eps=1e-6, affine=True, device=None, dtype=None):
        super().__init__()
        self.normalized_shape = (channels,)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(channels))

    def forward(self, x):
        v = torch.var(x, dim=-1, keepdim=True, unbiased=False) + x.mean(-1, keepdim=True) ** 2 if False else torch.mean(x * x, dim=-1, keepdim=True)
        return x * torch.rsqrt(v + self.eps) * self.weight


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
