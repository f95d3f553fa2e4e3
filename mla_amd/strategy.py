"""Training strategy counterpart (reference: training/strategies/base_strategy_mla.py:251-404 run_vla_training and
training/strategies/fsdp.py:176-310 run_setup / clip_grad_norm), driving mla_amd.fsdp.ShardedModel.

One call to ``train_step(batch)`` = one micro-step of the reference's hot loop: forward (bf16), backward and -- on the last
micro-batch of an accumulation window (every call when grad_accumulation_steps == 1, the shipped setting) -- global grad-norm
clip, AdamW step, LR-scheduler step. Logging / W&B / checkpoint I/O of the
reference's loop are out of scope; the loss dict keys are the reference's (base_strategy_mla.py:326-334).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.distributed as dist

from .fsdp import ShardedModel


def cosine_with_warmup(step: int, warmup: int, total: int) -> float:
    """transformers.optimization.get_cosine_schedule_with_warmup (optimization.py:144) multiplier."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


class FSDPStrategy:
    def __init__(self, vlm, device_id, stage: str = "finetune", epochs: int = 1, max_steps: Optional[int] = None,
                 global_batch_size: int = 8, per_device_batch_size: int = 8, learning_rate: float = 2e-5,
                 weight_decay: float = 0.0, max_grad_norm: float = 1.0, lr_scheduler_type: str = "constant",
                 warmup_ratio: float = 0.0, enable_gradient_checkpointing: bool = True,
                 enable_mixed_precision_training: bool = True, reduce_in_full_precision: bool = True,
                 repeated_diffusion_steps: int = 4, cast_forward_inputs: bool = True, local_ops=None, **_):
        self.vlm, self.stage = vlm, stage
        self.device = torch.device("cuda", device_id) if isinstance(device_id, int) else torch.device(device_id)
        if self.device.type == "cuda":
            # the reference does this in scripts/train.py:80 (torch.cuda.set_device(device_id)); libmla_hip.so launches go to the
            # current device's current stream, so the strategy pins it rather than trusting the caller
            torch.cuda.set_device(self.device)
        self.epochs, self.max_steps = epochs, max_steps
        self.global_batch_size, self.per_device_batch_size = global_batch_size, per_device_batch_size
        self.learning_rate, self.weight_decay, self.max_grad_norm = learning_rate, weight_decay, max_grad_norm
        self.lr_scheduler_type, self.warmup_ratio = lr_scheduler_type, warmup_ratio
        self.enable_gradient_checkpointing = enable_gradient_checkpointing
        self.repeated_diffusion_steps = repeated_diffusion_steps
        self.cast_forward_inputs = cast_forward_inputs
        if not enable_mixed_precision_training or not reduce_in_full_precision:
            raise NotImplementedError("only the shipped policy is built: bf16 parameters/compute, fp32 gradient reduction")
        self.local_ops = local_ops
        self.sharded: Optional[ShardedModel] = None
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        assert self.global_batch_size % (self.per_device_batch_size * self.world) == 0
        self.grad_accumulation_steps = self.global_batch_size // self.per_device_batch_size // self.world
        self.step = 0               # optimizer steps taken
        self._micro = 0             # micro-batches seen inside the current accumulation window
        self.num_training_steps = self.num_warmup_steps = 0

    def run_setup(self, n_train_examples: int = 0, run_dir=None) -> None:
        """fsdp.py:176-306: shard, (checkpointing), AdamW groups (no decay on <2-D / bias), LR schedule."""
        if self.enable_gradient_checkpointing:
            self.vlm.llm_backbone.enable_gradient_checkpointing()
        self.sharded = ShardedModel(self.vlm, self.vlm.get_fsdp_wrapping_policy(), self.device, ops=self.local_ops)
        n = math.ceil(max(n_train_examples, 1) / self.global_batch_size) * self.global_batch_size
        self.num_training_steps = self.max_steps if self.max_steps is not None else (n * self.epochs) // self.global_batch_size
        if self.lr_scheduler_type == "linear-warmup+cosine-decay":
            self.num_warmup_steps = int(self.num_training_steps * self.warmup_ratio)
        elif self.lr_scheduler_type == "constant":
            self.num_warmup_steps = 0
        else:
            raise ValueError(f"Learning Rate Schedule with type `{self.lr_scheduler_type}` is not supported!")
        if dist.is_available() and dist.is_initialized():
            dist.barrier()

    def current_lr(self) -> float:
        if self.lr_scheduler_type == "constant":
            return self.learning_rate
        return self.learning_rate * cosine_with_warmup(self.step, self.num_warmup_steps, self.num_training_steps)

    def _cast_inputs(self, batch: Dict) -> Dict:
        """torch FSDP casts every floating-point forward input to the compute dtype at the root
        (cast_root_forward_inputs default); reproduced at the module boundary."""
        def c(v):
            if torch.is_tensor(v):
                v = v.to(self.device, non_blocking=True)
                return v.to(torch.bfloat16) if (self.cast_forward_inputs and v.is_floating_point()) else v
            if isinstance(v, dict):
                return {k: c(x) for k, x in v.items()}
            return v
        return {k: c(v) for k, v in batch.items()}

    def train_step(self, batch: Dict) -> Dict[str, torch.Tensor]:
        """One micro-batch of the hot loop (base_strategy_mla.py:296-377). With grad_accumulation_steps == 1 (every shipped
        script) that is one optimizer step; otherwise the loss is divided by the window length (:365), gradients add up in the
        fp32 main_grad buffers, and only the last micro-batch of the window reduce-scatters, clips and steps (:370-377). The
        reference's FSDP reduces after every micro-batch; reducing the fp32 sum once is the same sum with fewer collectives."""
        sm = self.sharded
        acc = self.grad_accumulation_steps
        last = self._micro == acc - 1
        if self._micro == 0:
            sm.begin_step()
        sm.defer_reduce = not last
        self.vlm.train()
        b = self._cast_inputs(batch)
        loss_dict, _output = self.vlm(
            input_ids=b["input_ids"], attention_mask=b["attention_mask"], images=b["images"], next_images=b.get("next_images"),
            camera_name=b["camera_name"], point_cloud=b.get("point_cloud"), next_point_cloud=b.get("next_point_cloud"),
            tactile=b.get("tactile"), next_tactile=b.get("next_tactile"), labels=b["labels"], actions=b["actions"],
            proprio=b["proprio"], gripper_xyz=b.get("gripper_xyz"), action_masks=b.get("action_masks"), output_hidden_states=True,
            repeated_diffusion_steps=self.repeated_diffusion_steps, use_diff=True)
        (loss_dict["total_loss"] if acc == 1 else loss_dict["total_loss"] / acc).backward()
        if last:
            sm.finish_backward()
            self.clip_grad_norm()
            sm.optimizer_step(self.current_lr(), weight_decay=self.weight_decay)
            self.step += 1
            self._micro = 0
        else:
            sm.finish_micro_backward()
            self._micro += 1
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in loss_dict.items()}

    def synchronize(self) -> None:
        """Main stream waits for every outstanding side-stream all-gather (module forwards do this per unit on their own). Call it
        before reading parameter storage directly on the main stream; ``save_checkpoint`` / ``full_state_dict_fp32`` do."""
        if self.sharded is not None and self.sharded.on_gpu:
            self.sharded.wait_all()

    # ------------------------------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, run_dir, global_step: int, epoch: int, train_loss: Optional[float] = None, only_trainable: bool = True):
        """training/strategies/fsdp.py:100-141: gather the full fp32 state dict (every rank takes part in the gathers), split it
        by module key (``vlm.`` prefix dropped), rank 0 writes ``{"model": {mkey: OrderedDict}}`` to
        ``run_dir/checkpoints/step-XXXXXX-epoch-XX-loss=....pt``. Optimizer state is not saved (commented out in the reference).
        Returns the path (rank 0) or None."""
        from collections import OrderedDict
        from pathlib import Path
        assert self.sharded is not None, "save_checkpoint needs run_setup() first"
        # one unit at a time; only rank 0 keeps (host) copies -- FullStateDictConfig(offload_to_cpu=True, rank0_only=True)
        full = {k: v for k, v in self.sharded.iter_full_state_fp32(to_cpu_on_rank0=True) if v is not None}
        mkeys = list(self.vlm.trainable_module_keys if only_trainable else self.vlm.all_module_keys)
        model_state_dicts = {mkey: OrderedDict() for mkey in mkeys}
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if rank == 0:
            for key, val in self.vlm.state_dict().items():      # module order; buffers (BatchNorm statistics) come from here
                for mkey in mkeys:
                    if key.startswith(mkey + "."):
                        model_state_dicts[mkey][key[len(mkey) + 1:]] = full[key] if key in full else val.detach().cpu().clone()
        path = None
        if rank == 0:
            ckpt_dir = Path(run_dir) / "checkpoints"
            ckpt_dir.mkdir(parents=True, exist_ok=True)
            tag = "inf" if train_loss is None else f"{train_loss:.4f}"
            path = ckpt_dir / f"step-{global_step:06d}-epoch-{epoch:02d}-loss={tag}.pt"
            out = OrderedDict((k[4:] if k.startswith("vlm.") else k, v) for k, v in model_state_dicts.items())
            torch.save({"model": out}, path)
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        return path

    def clip_grad_norm(self):
        return self.sharded.grad_norm_and_clip(self.max_grad_norm)
