from timm.layers import Attention, Mlp, RmsNorm  # noqa: F401
import torch.nn as _nn


class Block(_nn.Module):           # names imported (never instantiated on the MLA path) by models/backbones/vision/*.py
    pass


class VisionTransformer(_nn.Module):
    pass


class LayerScale(_nn.Module):
    pass
