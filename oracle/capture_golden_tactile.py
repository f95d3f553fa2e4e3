"""Golden vectors for the tactile path (tactile tokens, TactileContrastiveLoss, TactileGenerationModule)
from the REAL reference -- build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden_tactile.py

Tiny MLA with use_pointcloud, use_contrastive, use_tactile and the tactile generation head (gen_image / gen_pointcloud off),
stage "post-training", dropout zeroed on the head (it only runs in train mode); fp32 (mode A) and bf16 autocast (mode C).
Writes tests/golden/mla_tiny_e2e_tactile.npz.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recipe, ref_import  # noqa: E402
from oracle.capture_golden import _Draws  # noqa: E402
from oracle.capture_golden_gen import zero_dropout  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
GEN = dict(use_generation=True, gen_image=False, use_roi=False, gen_pointcloud=False, gen_tactile=True)
SLICES = ("vlm.tactile_embedder.mlp.fc1.weight", "vlm.llm_backbone.llm.tactile_contrastive_loss_module.tactile_projection_head.2.weight",
          "vlm.llm_backbone.llm.tactile_contrastive_loss_module.image_projection_head.0.weight",
          "vlm.generation_manager.tactile_gen_module.output_head.weight",
          "vlm.generation_manager.tactile_gen_module.decoder.layers.1.multihead_attn.in_proj_weight")


def run(mode, R=2):
    mla = ref_import.build_reference_mla(recipe.TINY_LLAMA | {"vocab_size": recipe.TINY_LLAMA["vocab_size"] + 1}, recipe.TOKEN_SIZE,
                                         generation=GEN, use_tactile=True)
    shapes = {k: tuple(v.shape) for k, v in mla.state_dict().items()}
    mla.load_state_dict(recipe.make_state_dict(shapes), strict=True)
    mla.freeze_backbones("post-training")
    zero_dropout(mla.vlm.generation_manager)
    mla.train()
    batch, draws = recipe.make_batch(R=R, with_tactile=True)
    kw = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"], images=batch["images"],
              point_cloud=batch["point_cloud"], tactile=batch["tactile"], next_tactile=batch["next_tactile"], gripper_xyz=batch["gripper_xyz"],
              actions=batch["actions"], proprio=batch["proprio"], action_masks=batch["action_masks"], camera_name=batch["camera_name"],
              output_hidden_states=True, repeated_diffusion_steps=R, use_diff=True)
    import builtins
    _print = builtins.print
    builtins.print = lambda *a, **k: None
    try:
        if mode == "A":
            up = lambda m, a: tuple(x.float() if torch.is_tensor(x) and x.is_floating_point() else x for x in a)  # noqa: E731
            for emb in (mla.vlm.proprio_embedder, mla.vlm.x_embedder):
                emb.register_forward_pre_hook(up)
            with _Draws(draws, 2 * R):
                ld, out = mla(**kw)
        else:
            mla.to(torch.bfloat16)
            kw["images"] = {k: v.to(torch.bfloat16) for k, v in kw["images"].items()}
            for k in ("point_cloud", "actions", "proprio", "tactile", "next_tactile", "gripper_xyz"):
                kw[k] = kw[k].to(torch.bfloat16)
            with _Draws(draws, 2 * R), torch.autocast("cpu", dtype=torch.bfloat16):
                ld, out = mla(**kw)
        ld["total_loss"].float().backward()
    finally:
        builtins.print = _print
    grads = {k: p.grad for k, p in mla.named_parameters() if p.grad is not None}
    return shapes, ld, grads


def main():
    res = {}
    for mode in ("A", "C"):
        shapes, ld, grads = run(mode)
        f = lambda t: t.detach().float().numpy()  # noqa: E731
        for k, v in ld.items():
            res[f"{mode}_{k}"] = f(v)
        res[f"{mode}_gradnorms"] = np.array([float(grads[k].float().norm()) for k in sorted(grads)], dtype=np.float64)
        for k in sorted(grads):                        # round 6: an A and a C sample of EVERY parameter's gradient (recipe.grad_slice)
            res[f"{mode}_gs::{k}"] = f(recipe.grad_slice(grads[k]))
        for k in SLICES:
            g = grads[k]
            res[f"{mode}_grad::{k}"] = f(g.reshape(g.shape[0], -1)[:16, :64])
        if mode == "A":
            res["grad_names"] = np.array(sorted(grads))
            res["param_names"] = np.array(sorted(shapes))
            res["param_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes)])
    np.savez_compressed(os.path.join(OUT, "mla_tiny_e2e_tactile.npz"), **res)
    print("mla_tiny_e2e_tactile.npz:", {k: float(v) for k, v in res.items() if np.ndim(v) == 0})


if __name__ == "__main__":
    main()
