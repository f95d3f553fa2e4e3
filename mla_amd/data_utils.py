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

from .action_tokenizer import ActionTokenizer  # noqa: F401  (vla/action_tokenizer.py; the transform below calls it per action row)

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


# ------------------------------------------------------------------------------------------------- prompt / label construction
class PurePromptBuilder:
    """models/backbones/llm/prompting/base_prompter.py:27-79 (the builder `llama2-7b-pure` selects, llama2.py:81-83):
    alternating human / gpt turns; a human turn becomes ``"In: {msg}\\nOut: "``, a gpt turn ``"{msg}</s>"`` (no blank in front of an
    empty answer); ``<image>`` tags are dropped and the message stripped; the tokenizer adds BOS itself."""
    bos, eos = "<s>", "</s>"

    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        self.model_family, self.system_prompt = model_family, system_prompt
        self.prompt, self.turn_count = "", 0

    def _wrap(self, human: bool, msg: str) -> str:
        return f"In: {msg}\nOut: " if human else f"{msg}{self.eos}"

    def add_turn(self, role: str, message: str) -> str:
        human = self.turn_count % 2 == 0
        assert role == ("human" if human else "gpt")
        wrapped = self._wrap(human, message.replace("<image>", "").strip())
        self.prompt += wrapped
        self.turn_count += 1
        return wrapped

    def get_potential_prompt(self, message: str) -> str:
        return (self.prompt + self._wrap(True, message)).removeprefix(self.bos).rstrip()

    def get_prompt(self) -> str:
        return self.prompt.removeprefix(self.bos).rstrip()


@dataclass
class RLDSBatchTransform:
    """vla/datasets/datasets.py:30-185: one RLDS sample -> the per-sample dict the collator above consumes.

    * every camera frame (``image_primary`` -> ``front_image``, ``image_wrist_right/left``) goes through ``image_transform.preprocess(
      img, return_tensors="pt")["pixel_values"][0]`` (CLIPImageProcessor on the reference side; ``mla_amd.vision_tokenizer.
      ClipImagePreprocessor`` is the bit-exact GPU counterpart for the inference path) and gets a ones mask as 4th channel (:66-76);
      the next-frame target ``image_next_primary`` is transformed but carries no mask channel (:56-58);
    * tactile vectors: 65535 (sensor "no reading") -> 0, right ‖ left, / 100 (:79-97); point clouds are row 0 of the window (:106-110);
    * prompt: human "What action should the robot take to {instruction.lower()}?", gpt "<BOD><EOD>{action tokens}" (or "" without an
      action tokenizer), built with ``prompt_builder_fn("openvla")``, tokenised with special tokens (:112-141);
    * labels: a copy of the ids with everything but the last ``action_dim + 1`` (action tokens + EOS) positions ignored -- only the
      last position without an action tokenizer -- and the EOS position as well unless ``predict_stop_token`` (:143-163).
    ``image_transform`` may be any object with that ``preprocess`` signature; frames arrive as uint8 HWC arrays."""
    action_tokenizer: Optional[ActionTokenizer]
    base_tokenizer: object
    image_transform: object
    prompt_builder_fn: type
    predict_stop_token: bool = True
    use_pointcloud: bool = False
    use_tactile: bool = False

    def _frame(self, arr) -> torch.Tensor:
        from PIL import Image
        return self.image_transform.preprocess(Image.fromarray(arr), return_tensors="pt")["pixel_values"][0]

    @staticmethod
    def _tactile(obs, prefix: str) -> torch.Tensor:
        parts = []
        for side in ("right", "left"):
            t = torch.tensor(obs[f"{prefix}tactile_{side}"][0], dtype=torch.float32)
            parts.append(torch.where(t == 65535, torch.zeros((), dtype=t.dtype), t))
        return torch.cat(parts, dim=0) / 100.0

    def __call__(self, rlds_batch: Dict) -> Dict:
        obs = rlds_batch["observation"]
        action, proprio = rlds_batch["action"], obs["proprio"]                # the whole window (future-action chunks), :39-42
        front = self._frame(obs["image_primary"][0])
        mask = torch.ones(1, 672, 672)
        images = {"front_image": torch.cat([front, mask], dim=0)}
        next_image = self._frame(obs["image_next_primary"][0]) if "image_next_primary" in obs else None
        for key, name in (("image_wrist_right", "wrist_right_image"), ("image_wrist_left", "wrist_left_image")):
            if key in obs:
                images[name] = torch.cat([self._frame(obs[key][0]), mask], dim=0)
        tactile = next_tactile = gripper_xyz = None
        if self.use_tactile:
            tactile, next_tactile = self._tactile(obs, ""), self._tactile(obs, "next_")
            gripper_xyz = torch.tensor(obs["gripper_xyz"][0], dtype=torch.float32)
        pc = next_pc = None
        if self.use_pointcloud:
            pc = torch.tensor(obs["point_cloud"][0]).to(torch.float)
            next_pc = torch.tensor(obs["next_point_cloud"][0]).to(torch.float)
        lang = rlds_batch["task"]["language_instruction"].decode().lower()
        answer = "" if self.action_tokenizer is None else "<BOD><EOD>" + "".join(self.action_tokenizer(a) for a in action)
        builder = self.prompt_builder_fn("openvla")
        builder.add_turn("human", f"What action should the robot take to {lang}?")
        builder.add_turn("gpt", answer)
        input_ids = torch.tensor(self.base_tokenizer(builder.get_prompt(), add_special_tokens=True).input_ids)
        labels = input_ids.clone()
        action_t = torch.tensor(action, dtype=torch.float32)
        keep = 1 if self.action_tokenizer is None else len(action_t[0]) + 1
        labels[:-keep] = IGNORE_INDEX
        if not self.predict_stop_token:
            labels[-1] = IGNORE_INDEX
        return dict(images=images, point_cloud=pc, next_images=next_image, next_point_cloud=next_pc, tactile=tactile,
                    next_tactile=next_tactile, input_ids=input_ids, labels=labels, dataset_name=rlds_batch["dataset_name"],
                    actions=action_t, action_masks=torch.tensor(rlds_batch["action_mask"], dtype=torch.bool) if "action_mask" in rlds_batch else None,
                    proprio=torch.tensor(proprio, dtype=torch.float32), gripper_xyz=gripper_xyz)
