"""Debug (round 6): world-1 RCCL with the out-of-place AVG reduce-scatter vs the local gradient buffer, per parameter, after one step."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29572", MLA_FORCE_COLLECTIVES="1", MLA_FSDP_INPLACE_RS="0")
import torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from oracle import recipe
from tests_shapes import MLA_TINY_SHAPES
from mla_amd.backbones import LLaMa2LLMBackbone
from mla_amd.llama import LlamaConfig
from mla_amd.mla import MLA
from mla_amd.prismatic import PrismaticVLM
from mla_amd.strategy import FSDPStrategy
cfg = LlamaConfig(**recipe.TINY_LLAMA)
bb = LLaMa2LLMBackbone(config=cfg, pad_to_multiple_of=1)
vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=True, use_contrastive=True, use_generation=False)
m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=True, use_contrastive=True)
m.load_state_dict(recipe.make_state_dict(MLA_TINY_SHAPES))
m.freeze_backbones("finetune")
strat = FSDPStrategy(m, 0, global_batch_size=2, per_device_batch_size=2, learning_rate=1e-3, weight_decay=0.01, max_grad_norm=1e9,
                     lr_scheduler_type="constant", enable_gradient_checkpointing=False, repeated_diffusion_steps=2, cast_forward_inputs=False)
strat.run_setup(100)
sm = strat.sharded
batch, draws = recipe.make_batch(R=2)
m.vlm.vision_tower_3d.fps_starts_override = [draws["fps_start0"], draws["fps_start1"]]
orig = m.forward
m.forward = lambda **kw: orig(**kw, noise=draws["noise"].to(dev), timestep=draws["timestep"].to(dev))
for step in range(2):
    strat.train_step(batch)
    strat.synchronize(); torch.cuda.synchronize()
    for u in sm.units:
        if not u.trainable:
            continue
        same = torch.equal(u.gshard, u.grad32)
        print(f"step {step} unit {u.name:60s} rs_from_hook={u.rs_event is not None} gshard == grad32: {same}")
        if not same:
            for n, p, o in u.params:
                if p.requires_grad:
                    a, b = u.gshard[o:o + p.numel()], u.grad32[o:o + p.numel()]
                    if not torch.equal(a, b):
                        print(f"      {n}: |gshard| {float(a.norm()):.4e} |grad32| {float(b.norm()):.4e} |diff| {float((a - b).norm()):.4e}")
dist.destroy_process_group()
