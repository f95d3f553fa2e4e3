"""Golden vectors for the image generation head with use_roi=True (ROI dilation, mask tokens, translation warp, alpha blend,
background loss) from the REAL reference -- build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden_roi.py

Dropout zeroed; fp32 (A) and bf16 autocast (C). Writes tests/golden/generation_roi.npz.
"""
import contextlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recipe, ref_import  # noqa: E402
from oracle.capture_golden_gen import gen_inputs, zero_dropout  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
PFX = "vlm.generation_manager."


def roi_inputs(B=4):
    g = recipe._gen("gen.roi")
    feats = recipe.det_randn("gen.img_feats", (B, 256, recipe.TOKEN_SIZE))
    idx = torch.randint(3, 12, (B, 24, 2), generator=g)           # 24 projected point centres per sample, clustered
    return feats, idx


def main():
    ref_import.setup()
    from models.mla.generation import MultimodalGenerationManager, create_roi_mask_from_indices, images_to_patches
    from models.vlm.prismatic import PrismaticVLM
    gt = recipe.GEN_TINY
    res = {}
    for mode in ("A", "C"):
        mgr = MultimodalGenerationManager(token_size=recipe.TOKEN_SIZE, use_image_generation=True, num_image_gen_queries=gt["num_image_gen_queries"],
                                          image_decoder_layers=gt["image_decoder_layers"], image_decoder_heads=gt["image_decoder_heads"],
                                          image_patch_size=42, use_roi=True, roi_dilation_kernel_size=3, use_pointcloud_generation=False)
        shapes = {k: tuple(v.shape) for k, v in mgr.state_dict().items()}
        sd = {k: recipe.det_weight(PFX + k, s) for k, s in shapes.items()}
        # heads that start near zero in training would make alpha / offsets trivial: give them ordinary weights, centred biases
        sd["image_gen_module.mae_alpha_head.bias"] = torch.zeros(1)
        mgr.load_state_dict(sd, strict=True)
        zero_dropout(mgr)
        mgr.train()
        hidden, curr, nxt, _ = gen_inputs()
        feats, idx = roi_inputs()
        if mode == "C":
            mgr.to(torch.bfloat16)
            hidden, curr, nxt, feats = hidden.bfloat16(), curr.bfloat16(), nxt.bfloat16(), feats.bfloat16()
        hidden.requires_grad_(True)
        feats.requires_grad_(True)
        stub = types.SimpleNamespace(gen_image=True, gen_pointcloud=False, gen_tactile=False, generation_manager=mgr)
        ctx = torch.autocast("cpu", dtype=torch.bfloat16) if mode == "C" else contextlib.nullcontext()
        with ctx:
            outs = mgr(llm_hidden_states=hidden, current_image_features=feats, current_images_patches=images_to_patches(curr[:, :3], 42),
                       current_point_cloud=None, roi_mask_2d=create_roi_mask_from_indices(idx))
            losses = PrismaticVLM.compute_generation_losses(stub, outs, next_images=nxt)
        losses["image_gen_loss"].float().backward()
        f = lambda t: t.detach().float().numpy()  # noqa: E731
        for k in ("image_gen_loss", "image_roi_generation_loss", "bg_consistency_loss", "delta_magnitude_reward"):
            res[f"{mode}_{k}"] = f(losses[k])
        res[f"{mode}_roi_mask"] = outs["generation_roi_mask"].numpy()
        res[f"{mode}_alpha"] = f(outs["alpha_all"])
        res[f"{mode}_offset"] = f(outs["offset_all"])
        res[f"{mode}_generation_slice"] = f(outs["image_generation"][:, ::16, ::97])
        res[f"{mode}_hidden_grad"] = f(hidden.grad)
        res[f"{mode}_feats_grad"] = f(feats.grad)
        grads = {k: p.grad for k, p in mgr.named_parameters() if p.grad is not None}
        res[f"{mode}_gradnorms"] = np.array([float(grads[k].float().norm()) for k in sorted(grads)], dtype=np.float64)
        for k in ("image_gen_module.mae_alpha_head.weight", "image_gen_module.mae_offset_head.weight", "image_gen_module.mae_delta_head.weight"):
            res[f"{mode}_grad::{k}"] = f(grads[k][:16, :64])
        if mode == "A":
            res["grad_names"] = np.array(sorted(grads))
            res["param_names"] = np.array(sorted(shapes))
            res["param_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes)])
    np.savez_compressed(os.path.join(OUT, "generation_roi.npz"), **res)
    print("generation_roi.npz:", {k: float(v) for k, v in res.items() if np.ndim(v) == 0}, "roi patches", int(res["A_roi_mask"].sum()), "of", res["A_roi_mask"].size)
    print(" alpha range", res["A_alpha"].min(), res["A_alpha"].max(), "offset range", res["A_offset"].min(), res["A_offset"].max())


if __name__ == "__main__":
    main()
