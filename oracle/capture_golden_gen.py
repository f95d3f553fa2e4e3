"""Golden vectors for the post-training generation heads, from the REAL reference classes
(models/mla/generation/models.py, PrismaticVLM.compute_generation_losses) imported from /root/reference -- build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden_gen.py

Dropout / attention-dropout / DropPath are set to p = 0 on the reference modules (they are stochastic in train mode and
the heads only run in train mode, prismatic.py:1075); BatchNorm stays on batch statistics. Weights and inputs are functions of
names + seeds (oracle/recipe.py); only outputs are stored.
Writes tests/golden/generation.npz (heads alone, fp32 "A" and bf16-autocast "C") and tests/golden/mla_tiny_e2e_gen.npz
(whole tiny MLA in post-training mode, fp32).
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recipe, ref_import  # noqa: E402
from oracle.capture_golden import _Draws  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
PFX = "vlm.generation_manager."


def zero_dropout(mod):
    import torch.nn as nn
    for m in mod.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if isinstance(m, nn.MultiheadAttention):
            m.dropout = 0.0
        if type(m).__name__ == "DropPath":
            m.drop_prob = 0.0


def gen_inputs(B=4, S=45):
    hidden = recipe.det_randn("gen.hidden", (B, S, recipe.TOKEN_SIZE))
    curr = recipe.det_randn("gen.curr", (B, 4, 672, 672))
    nxt = recipe.det_randn("gen.next", (B, 3, 672, 672))
    lo, hi = torch.tensor([0.0, -0.4, 0.75]), torch.tensor([0.6, 0.4, 1.25])
    npc = lo + (hi - lo) * torch.rand(B, 1024, 3, generator=recipe._gen("gen.next_pc"))
    return hidden, curr, nxt, npc


def capture_heads():
    ref_import.setup()
    from models.mla.generation import MultimodalGenerationManager, images_to_patches
    from models.vlm.prismatic import PrismaticVLM
    g = recipe.GEN_TINY
    res = {}
    for mode in ("A", "C"):
        mgr = MultimodalGenerationManager(
            token_size=recipe.TOKEN_SIZE, use_image_generation=True, num_image_gen_queries=g["num_image_gen_queries"],
            image_decoder_layers=g["image_decoder_layers"], image_decoder_heads=g["image_decoder_heads"], image_patch_size=42,
            use_roi=False, use_pointcloud_generation=True, pointcloud_trans_dim=g["pointcloud_trans_dim"],
            pointcloud_decoder_layers=g["pointcloud_decoder_layers"], pointcloud_decoder_heads=g["pointcloud_decoder_heads"],
            pointcloud_group_size=g["pointcloud_group_size"], pointcloud_num_groups=g["pointcloud_num_groups"])
        shapes = {k: tuple(v.shape) for k, v in mgr.state_dict().items()}
        mgr.load_state_dict({k: recipe.det_weight(PFX + k, s) for k, s in shapes.items()}, strict=True)
        zero_dropout(mgr)
        mgr.train()
        hidden, curr, nxt, npc = gen_inputs()
        if mode == "C":
            mgr.to(torch.bfloat16)
            hidden, curr, nxt = hidden.bfloat16(), curr.bfloat16(), nxt.bfloat16()
        hidden.requires_grad_(True)
        stub = types.SimpleNamespace(gen_image=True, gen_pointcloud=True, gen_tactile=False, generation_manager=mgr)
        import contextlib
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if mode == "C" else contextlib.nullcontext()
        with ctx:
            outs = mgr(llm_hidden_states=hidden, current_image_features=torch.zeros(hidden.shape[0], 256, recipe.TOKEN_SIZE, dtype=hidden.dtype),
                       current_images_patches=images_to_patches(curr[:, :3], 42), current_point_cloud=None,
                       roi_mask_2d=torch.ones(hidden.shape[0], 16, 16, dtype=torch.bool))
            losses = PrismaticVLM.compute_generation_losses(stub, outs, next_images=nxt, next_point_cloud=npc.to(hidden.dtype) if mode == "C" else npc)
        (losses["image_gen_loss"].float() + losses["point_cloud_gen_loss"].float()).backward()
        f = lambda t: t.detach().float().numpy()  # noqa: E731
        for k in ("image_gen_loss", "point_cloud_gen_loss", "image_roi_generation_loss", "delta_magnitude_reward", "total_generation_loss"):
            res[f"{mode}_{k}"] = f(losses[k])
        res[f"{mode}_delta_slice"] = f(outs["delta_all"][:, ::16, ::97])
        res[f"{mode}_image_generation_slice"] = f(outs["image_generation"][:, ::16, ::97])
        res[f"{mode}_points"] = f(outs["pointcloud_coord_generation"])
        res[f"{mode}_hidden_grad"] = f(hidden.grad)
        grads = {k: p.grad for k, p in mgr.named_parameters() if p.grad is not None}
        res[f"{mode}_gradnorms"] = np.array([float(grads[k].float().norm()) for k in sorted(grads)], dtype=np.float64)
        if mode == "A":
            res["grad_names"] = np.array(sorted(grads))
            res["param_names"] = np.array(sorted(shapes))
            res["param_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes)])
            res["no_grad_names"] = np.array(sorted(k for k, p in mgr.named_parameters() if p.grad is None))
        # gradient slices in BOTH modes (round 5: mode C added so that the per-slice bounds of the GPU test are the yardstick too)
        for k in ("image_gen_module.mae_delta_head.weight", "image_gen_module.intent_decoder.layers.0.multihead_attn.in_proj_weight",
                  "image_gen_module.mae_decoder.layers.1.linear1.weight", "pointcloud_gen_module.seq_to_patch.weight",
                  "pointcloud_gen_module.decoder_blocks.0.attn.in_proj_weight", "pointcloud_gen_module.future_predictor.0.weight"):
            res[f"{mode}_grad::{k}"] = f(grads[k].reshape(grads[k].shape[0], -1)[:16, :64])
        for k in ("image_gen_module.image_gen_queries", "image_gen_module.mae_mask_token", "image_gen_module.mae_patch_norm.weight",
                  "pointcloud_gen_module.future_predictor.1.weight", "pointcloud_gen_module.future_predictor.1.bias",
                  "image_gen_module.mae_decoder.layers.0.self_attn.in_proj_bias"):
            res[f"{mode}_grad::{k}"] = f(grads[k].reshape(-1)[:256])
        bn = mgr.pointcloud_gen_module.future_predictor[1]
        res[f"{mode}_bn_running_mean"], res[f"{mode}_bn_running_var"] = f(bn.running_mean), f(bn.running_var)
    np.savez_compressed(os.path.join(OUT, "generation.npz"), **res)
    print("generation.npz:", {k: float(res[k]) for k in res if k.endswith("_loss")})
    print("  params without gradient:", list(res["no_grad_names"]))


def _run_e2e_gen(mode, R=2):
    """mode "A": fp32 module, fp32 inputs (the plain import); "C": model.to(bf16) + autocast(cpu, bf16) with bf16 float inputs -- what
    FSDP mixed precision feeds the reference on the GPU; same recipe as capture_golden.run_reference."""
    g = recipe.GEN_TINY
    mla = ref_import.build_reference_mla(recipe.TINY_LLAMA | {"vocab_size": recipe.TINY_LLAMA["vocab_size"] + 1}, recipe.TOKEN_SIZE,
                                         generation=dict(use_generation=True, gen_image=True, use_roi=False, gen_pointcloud=True,
                                                         gen_tactile=False, **g))
    shapes = {k: tuple(v.shape) for k, v in mla.state_dict().items()}
    mla.load_state_dict(recipe.make_state_dict(shapes), strict=True)
    mla.freeze_backbones("post-training")
    zero_dropout(mla.vlm.generation_manager)
    mla.train()
    batch, draws = recipe.make_batch(R=R, with_next=True)
    kw = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"], images=batch["images"],
              next_images=batch["next_images"], point_cloud=batch["point_cloud"], next_point_cloud=batch["next_point_cloud"],
              actions=batch["actions"], proprio=batch["proprio"], action_masks=batch["action_masks"], camera_name=batch["camera_name"],
              gripper_xyz=None, output_hidden_states=True, repeated_diffusion_steps=R, use_diff=True)
    import builtins
    _print = builtins.print
    builtins.print = lambda *a, **k: None
    try:
        if mode == "A":
            up = lambda m, a: tuple(x.float() if torch.is_tensor(x) and x.is_floating_point() else x for x in a)  # noqa: E731
            mla.vlm.proprio_embedder.register_forward_pre_hook(up)
            mla.vlm.x_embedder.register_forward_pre_hook(up)
            with _Draws(draws, 2 * R):
                loss_dict, out = mla(**kw)
        else:
            mla.to(torch.bfloat16)
            kw["images"] = {k: v.to(torch.bfloat16) for k, v in kw["images"].items()}
            for k in ("point_cloud", "actions", "proprio", "next_images", "next_point_cloud"):
                kw[k] = kw[k].to(torch.bfloat16)
            with _Draws(draws, 2 * R), torch.autocast("cpu", dtype=torch.bfloat16):
                loss_dict, out = mla(**kw)
        loss_dict["total_loss"].float().backward()
    finally:
        builtins.print = _print
    grads = {k: p.grad for k, p in mla.named_parameters() if p.grad is not None}
    return shapes, loss_dict, grads


def capture_e2e_gen(R=2):
    res = {}
    f = lambda t: t.detach().float().numpy()  # noqa: E731
    for mode in ("A", "C"):
        shapes, loss_dict, grads = _run_e2e_gen(mode, R)
        res.update({f"{mode}_{k}": f(v) for k, v in loss_dict.items()})
        if mode == "A":
            res["grad_names"] = np.array(sorted(grads))
            res["param_names"] = np.array(sorted(shapes))
            res["param_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes)])
        res[f"{mode}_gradnorms"] = np.array([float(grads[k].float().norm()) if k in grads else 0.0 for k in res["grad_names"]], dtype=np.float64)
        for k in sorted(grads):                        # round 6: an A and a C sample of EVERY parameter's gradient (recipe.grad_slice)
            res[f"{mode}_gs::{k}"] = f(recipe.grad_slice(grads[k]))
    np.savez_compressed(os.path.join(OUT, "mla_tiny_e2e_gen.npz"), **res)
    print("mla_tiny_e2e_gen.npz:", {k: float(v) for k, v in res.items() if k[:2] in ("A_", "C_") and np.ndim(v) == 0})


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    capture_heads()
    capture_e2e_gen()
