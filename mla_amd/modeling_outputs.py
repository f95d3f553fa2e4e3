"""CausalLMOutputWithPast with the MLA extension fields (reference: transformers/modeling_outputs.py:706-713)."""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch


@dataclass
class CausalLMOutputWithPast:
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    img_pc_contrastive_loss: Optional[torch.Tensor] = None
    tactile_contrastive_loss: Optional[torch.Tensor] = None
    all_logits_for_action: Optional[torch.Tensor] = None
    past_key_values: Optional[Tuple] = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Optional[Tuple[torch.Tensor, ...]] = None

    def __getitem__(self, k):
        if isinstance(k, str):
            return getattr(self, k)
        return tuple(v for v in (self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions)
                     if v is not None)[k]
