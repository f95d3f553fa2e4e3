"""Batch collation on the caller side of the path (reference: util/data_utils.py:88-196, PaddedCollatorForActionPrediction).

Turns the per-sample dicts of the dataset into the keyword arguments of ``MLA.forward``: right-padded ``input_ids`` (pad id) / ``labels``
(-100) truncated to ``model_max_length``, ``attention_mask = input_ids != pad``, everything else stacked; optional fields (tactile,
next_tactile, action_masks) become None when the first sample has none; point clouds pass through un-stacked when they are not tensors.
Pure host logic (torch CPU tensors in, torch CPU tensors out) -- the HIP path starts at ``FSDPStrategy.train_step``."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import torch
from torch.nn.utils.rnn import pad_sequence

IGNORE_INDEX = -100


def _stack_views(items):
    """list of tensors -> stacked tensor; list of {view: tensor} -> {view: stacked tensor}."""
    first = items[0]
    if isinstance(first, dict):
        return {k: torch.stack([it[k] for it in items]) for k in first}
    if torch.is_tensor(first):
        return torch.stack(list(items))
    raise ValueError(f"Unsupported image container type = {type(first)}")


def _stack_optional(instances, key) -> Optional[torch.Tensor]:
    if key not in instances[0] or instances[0][key] is None:
        return None
    return torch.stack([inst[key] for inst in instances])


@dataclass
class PaddedCollatorForActionPrediction:
    model_max_length: int
    pad_token_id: int
    padding_side: str = "right"
    pixel_values_dtype: torch.dtype = torch.float32

    def __call__(self, instances: Sequence[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
        assert self.padding_side == "right", f"Invalid Tokenizer `{self.padding_side = }`"
        ids = pad_sequence([inst["input_ids"] for inst in instances], batch_first=True, padding_value=self.pad_token_id)
        labels = pad_sequence([inst["labels"] for inst in instances], batch_first=True, padding_value=IGNORE_INDEX)
        ids, labels = ids[:, : self.model_max_length], labels[:, : self.model_max_length]
        clouds = {}
        for key in ("point_cloud", "next_point_cloud"):
            vals = [inst[key] for inst in instances]
            clouds[key] = torch.stack(vals) if torch.is_tensor(vals[0]) else vals
        tactile = _stack_optional(instances, "tactile")
        out = dict(images=_stack_views([inst["images"] for inst in instances]),
                   next_images=_stack_views([inst["next_images"] for inst in instances]),
                   point_cloud=clouds["point_cloud"], next_point_cloud=clouds["next_point_cloud"],
                   tactile=tactile, next_tactile=_stack_optional(instances, "next_tactile"),
                   input_ids=ids, attention_mask=ids.ne(self.pad_token_id), labels=labels,
                   actions=torch.stack([inst["actions"] for inst in instances]),
                   action_masks=_stack_optional(instances, "action_masks"),
                   proprio=torch.stack([inst["proprio"] for inst in instances]),
                   gripper_xyz=torch.stack([inst["gripper_xyz"] for inst in instances]) if tactile is not None else None)
        if "dataset_name" in instances[0]:
            out["dataset_names"] = [inst["dataset_name"] for inst in instances]
        return out
