"""MLPProjector (reference: util/nn_utils.py:21-34), trainable projector_3d."""
import torch.nn as nn

from . import hip, ops
from .llama import Linear


class MLPProjector(nn.Module):
    def __init__(self, vision_dim: int, llm_dim: int, mlp_type: str = "gelu-mlp") -> None:
        super().__init__()
        if mlp_type != "gelu-mlp":
            raise ValueError(f"Projector with `{mlp_type = }` is not supported!")
        self.projector = nn.Sequential(Linear(vision_dim, llm_dim, bias=True), nn.GELU(), Linear(llm_dim, llm_dim, bias=True))

    def forward(self, img_patches):
        x = self.projector[0](img_patches)
        x = ops.act(x, hip.ACT_GELU_ERF)
        return self.projector[2](x)
