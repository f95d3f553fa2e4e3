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
                 repeated_diffusion_steps: int = 4, cast_forward_inputs: bool = True, local_ops=None,
                 mixed_precision_dtype: torch.dtype = torch.bfloat16, worker_init_fn=None, requires_cliploss: bool = False,
                 sharding_strategy: str = "full-shard", **_):
        """Constructor arguments of the reference's FSDPStrategy (training/strategies/fsdp.py:47-96) -- get_train_strategy passes
        them through unchanged. `sharding_strategy` "full-shard" and "shard-grad-op" run the same schedule here: bf16 weights stay
        replicated between the optimizer step and the next use (13.5 GB of 288 GB), master weights / gradients / AdamW moments are
        sharded 1/world (see mla_amd/fsdp.py)."""
        if mixed_precision_dtype != torch.bfloat16:
            raise NotImplementedError("the HIP path computes in bfloat16 (the reference's shipped mixed_precision_dtype)")
        if sharding_strategy not in ("full-shard", "shard-grad-op"):
            raise ValueError(f"FSDP Sharding Strategy {sharding_strategy} is not supported!")
        self.worker_init_fn, self.sharding_strategy = worker_init_fn, sharding_strategy
        # One layout for both names (deviation 12): torch FSDP's FULL_SHARD frees the gathered parameters after
        # each unit's forward / backward, SHARD_GRAD_OP keeps them until the backward is over; here the full bf16 replica (12.9 GiB at 7B)
        # and the full fp32 gradient buffer (25.8 GiB) stay resident on every rank and only masters + AdamW moments + the reduced
        # gradient shard are 1/world. Results are identical; say so once instead of silently treating the names as synonyms.
        import logging
        logging.getLogger(__name__).info(
            "FSDPStrategy(sharding_strategy=%r): 'full-shard' and 'shard-grad-op' select the SAME layout in this build -- bf16 weights "
            "replicated and resident, fp32 gradients reduced into 1/world shards, fp32 masters and AdamW moments sharded 1/world",
            sharding_strategy)
        self.last_lr = learning_rate
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

    def train_step(self, batch: Dict, camera_name: Optional[str] = None, use_pointcloud: Optional[bool] = None,
                   use_tactile: Optional[bool] = None, use_generation: Optional[bool] = None, gen_image: bool = True,
                   gen_pointcloud: bool = True, gen_tactile: bool = True,
                   repeated_diffusion_steps: Optional[int] = None) -> Dict[str, torch.Tensor]:
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
        # which optional streams reach the model: the reference's loop decides from its flags (base_strategy_mla.py:306-322);
        # a flag left at None means "whatever the batch carries"
        pc_on = ("point_cloud" in b) if use_pointcloud is None else use_pointcloud
        tac_on = ("tactile" in b) if use_tactile is None else use_tactile
        gen_on = True if use_generation is None else use_generation
        loss_dict, _output = self.vlm(
            input_ids=b["input_ids"], attention_mask=b["attention_mask"], images=b["images"],
            next_images=b.get("next_images") if (gen_on and gen_image) else None,
            camera_name=camera_name if camera_name is not None else b["camera_name"],
            point_cloud=b.get("point_cloud") if pc_on else None,
            next_point_cloud=b.get("next_point_cloud") if (pc_on and gen_on and gen_pointcloud) else None,
            tactile=b.get("tactile") if tac_on else None,
            next_tactile=b.get("next_tactile") if (tac_on and gen_on and gen_tactile) else None,
            labels=b["labels"], actions=b["actions"], proprio=b["proprio"], gripper_xyz=b.get("gripper_xyz"),
            action_masks=b.get("action_masks"), output_hidden_states=True,
            repeated_diffusion_steps=self.repeated_diffusion_steps if repeated_diffusion_steps is None else repeated_diffusion_steps,
            use_diff=True)
        (loss_dict["total_loss"] if acc == 1 else loss_dict["total_loss"] / acc).backward()
        if last:
            sm.finish_backward()
            self.clip_grad_norm()
            self.last_lr = self.current_lr()
            sm.optimizer_step(self.last_lr, weight_decay=self.weight_decay)
            self.step += 1
            self._micro = 0
        else:
            sm.finish_micro_backward()
            self._micro += 1
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in loss_dict.items()}

    def run_vla_training(self, vla_dataset, collator, metrics, save_interval: int = 2500, save_full_model: bool = True,
                         use_diff: bool = False, use_pointcloud: bool = False, use_tactile: bool = False,
                         use_contrastive: bool = False, camera_name: str = "", use_generation: bool = False,
                         gen_image: bool = False, gen_pointcloud: bool = False, gen_tactile: bool = False,
                         repeated_diffusion_steps=4) -> None:
        """training/strategies/base_strategy_mla.py:251-404 with the reference's signature, so scripts/train.py:405-420 calls it
        unchanged. `metrics` is the reference's VLAMetrics (duck-typed: get_status / commit / push / global_step / run_dir).
        Same control flow: an "infinite" DataLoader over the IterableDataset (batch = per_device_batch_size, the given collator,
        num_workers 0), one `train_step` per batch, metric commits with the reference's seven loss keys, an optimizer step every
        grad_accumulation_steps batches followed by the epoch / lr / step-time commit and `metrics.push()`, a checkpoint at
        max_steps or at every `save_interval`-th epoch boundary, termination after `epochs` passes. Not reproduced: tqdm, EMA of the
        (absent) DiT action model, and the non-diffusion branch (dead in the reference: SURVEY Appendix A #15)."""
        from torch.utils.data import DataLoader, IterableDataset
        assert isinstance(vla_dataset, IterableDataset), "VLA training expects an IterableDataset!"
        if not use_diff:
            raise NotImplementedError("only the use_diff=True branch of MLA.forward is alive in the reference (model_mla.py:236-275)")
        dataloader = DataLoader(vla_dataset, batch_size=self.per_device_batch_size, sampler=None, collate_fn=collator, num_workers=0,
                                worker_init_fn=self.worker_init_fn)
        steps_per_epoch = math.ceil(len(dataloader) / self.world / self.grad_accumulation_steps)
        metrics.get_status()
        self.vlm.train()
        for train_idx, batch in enumerate(dataloader):
            ld = self.train_step(batch, camera_name=camera_name, use_pointcloud=use_pointcloud, use_tactile=use_tactile,
                                 use_generation=use_generation, gen_image=gen_image, gen_pointcloud=gen_pointcloud,
                                 gen_tactile=gen_tactile, repeated_diffusion_steps=repeated_diffusion_steps)
            metrics.commit(loss=ld["total_loss"], img_pc_contrastive_loss=ld["img_pc_contrastive_loss"],
                           tactile_contrastive_loss=ld["tactile_contrastive_loss"], diff_loss=ld["diff_loss"],
                           image_gen_loss=ld["image_gen_loss"], point_cloud_gen_loss=ld["point_cloud_gen_loss"],
                           tactile_gen_loss=ld["tactile_gen_loss"])
            metrics.commit(loss=ld["total_loss"])
            if (train_idx + 1) % self.grad_accumulation_steps != 0:
                continue
            div = (len(vla_dataset) // self.global_batch_size) or 1
            epoch = (metrics.global_step + 1) // div
            # the reference commits lr_scheduler.get_last_lr()[0] AFTER lr_scheduler.step() (base_strategy_mla.py:375-377), i.e. the rate
            # the NEXT optimizer step will use; train_step has already advanced self.step, so current_lr() is that rate
            metrics.commit(update_step_time=True, global_step=metrics.global_step + 1, epoch=epoch, lr=self.current_lr())
            metrics.push()
            if (self.max_steps is not None and metrics.global_step >= self.max_steps) or (
                    metrics.global_step % steps_per_epoch == 0 and epoch % save_interval == 0):
                self.save_checkpoint(metrics.run_dir, metrics.global_step, epoch, float(ld["total_loss"]),
                                     only_trainable=not save_full_model)
            if metrics.global_step >= self.epochs * steps_per_epoch:
                return

    def load_optimizer_and_scheduler(self, checkpoint_path) -> None:
        """training/strategies/fsdp.py:161-174: the reference looks for `<checkpoint stem>-optimizer.pt`, warns and returns when it
        is absent -- which it always is, because the write side is commented out (fsdp.py:143-159) here as there."""
        import warnings
        from pathlib import Path
        cp = Path(checkpoint_path)
        opt = cp.with_name(cp.stem + "-optimizer.pt")
        if not opt.exists():
            warnings.warn(f"Optimizer checkpoint not found at {opt}!")
            return
        raise NotImplementedError("optimizer state files are never written by the reference's save_checkpoint (fsdp.py:143-159)")

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


# training/materialize.py:16-68 -- the registry and factory scripts/train.py:367-385 uses
TRAIN_STRATEGIES = {
    "fsdp-shard-grad-op": {"cls": FSDPStrategy, "kwargs": {"sharding_strategy": "shard-grad-op"}},
    "fsdp-full-shard": {"cls": FSDPStrategy, "kwargs": {"sharding_strategy": "full-shard"}},
}


def get_train_strategy(train_strategy: str, vlm, device_id: int, stage: str, epochs: int, max_steps: Optional[int],
                       global_batch_size: int, per_device_batch_size: int, learning_rate: float, weight_decay: float,
                       max_grad_norm: float, lr_scheduler_type: str, warmup_ratio: float,
                       enable_gradient_checkpointing: bool = True, enable_mixed_precision_training: bool = True,
                       reduce_in_full_precision: bool = False, mixed_precision_dtype: torch.dtype = torch.bfloat16,
                       worker_init_fn=None, requires_cliploss: bool = False) -> FSDPStrategy:
    """Same signature, defaults and error as training/materialize.py:22-68. NB the factory's default
    `reduce_in_full_precision=False` meets the strategy's fp32-only reduction: scripts/train.py always passes the VLAConfig value
    (True, conf/vla.py:55); a caller relying on the bf16-reduction default gets the NotImplementedError of FSDPStrategy.
    "fsdp-shard-grad-op" and "fsdp-full-shard" (training/strategies/fsdp.py:88-93) both resolve to the one layout this build has:
    resident bf16 replica + full fp32 gradient buffer per rank, masters / moments / reduced gradients sharded 1/world (38.8 GiB of
    un-sharded state per rank at every world size) -- the strategy logs that at INFO."""
    if train_strategy not in TRAIN_STRATEGIES:
        raise ValueError(f"Train Strategy `{train_strategy}` is not supported!")
    cfg = TRAIN_STRATEGIES[train_strategy]
    return cfg["cls"](vlm=vlm, device_id=device_id, stage=stage, epochs=epochs, max_steps=max_steps,
                      global_batch_size=global_batch_size, per_device_batch_size=per_device_batch_size, learning_rate=learning_rate,
                      weight_decay=weight_decay, max_grad_norm=max_grad_norm, lr_scheduler_type=lr_scheduler_type,
                      warmup_ratio=warmup_ratio, enable_gradient_checkpointing=enable_gradient_checkpointing,
                      enable_mixed_precision_training=enable_mixed_precision_training,
                      reduce_in_full_precision=reduce_in_full_precision, mixed_precision_dtype=mixed_precision_dtype,
                      worker_init_fn=worker_init_fn, requires_cliploss=requires_cliploss, **cfg["kwargs"])
