"""Golden vectors for stage "pretrain" (trainable vision tokenizer; BASELINE configs[4] shape: use_pointcloud=False) from the REAL
reference -- build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden_pretrain.py          # -> mla_tiny_e2e_pretrain.npz
    PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden_pretrain.py --pc     # point tower trained too -> mla_tiny_e2e_pretrain_pc.npz
    PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden_pretrain.py --eq R   # unpadded batch (equal text lengths), R diffusion repeats
                                                                               #   -> mla_tiny_e2e_pretrain_eq.npz: the fixture of the
                                                                               #   shared-prefix forward (round 6), which needs no padding

Tiny MLA (recipe weights), freeze_backbones("pretrain"), one forward/backward in fp32 (mode A) and bf16 autocast (mode C):
losses and every gradient norm, plus slices of the vision-tower gradients. Writes tests/golden/mla_tiny_e2e_pretrain.npz.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recipe, ref_import  # noqa: E402
from oracle.capture_golden import _Draws  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SLICES = ("vlm.vision_tower_2d.patch_embedding.weight", "vlm.vision_tower_2d.local_attention.q.1.weight",
          "vlm.vision_tower_2d.local_attention.kv.1.weight", "vlm.vision_tower_2d.local_attention.proj.weight")
VECS = ("vlm.vision_tower_2d.local_attention.q.0.weight", "vlm.vision_tower_2d.local_attention.kv.0.bias",
        "vlm.vision_tower_2d.local_attention.proj.bias")


PC_SLICES = ("vlm.vision_tower_3d.patch_embed.EncP.raw_point_embed.net.0.weight",
             "vlm.vision_tower_3d.patch_embed.EncP.LGA_list.0.linear2.0.net1.0.weight",
             "vlm.vision_tower_3d.patch_embed.EncP.LGA_list.0.linear2.1.net2.0.weight",
             "vlm.vision_tower_3d.patch_embed.EncP.LGA_list.1.linear2.0.net1.0.weight", "vlm.vision_tower_3d.proj.weight")
PC_VECS = ("vlm.vision_tower_3d.patch_embed.EncP.raw_point_embed.net.1.weight",
           "vlm.vision_tower_3d.patch_embed.EncP.LGA_list.0.linear2.0.net1.1.bias",
           "vlm.vision_tower_3d.patch_embed.EncP.LGA_list.1.linear2.0.net2.1.weight",
           "vlm.vision_tower_3d.patch_embed.EncP.LGA_list.1.linear2.0.net2.0.bias", "vlm.vision_tower_3d.proj.bias")


def run(mode, R=2, pc=False, ragged=True):
    mla = ref_import.build_reference_mla(recipe.TINY_LLAMA | {"vocab_size": recipe.TINY_LLAMA["vocab_size"] + 1}, recipe.TOKEN_SIZE,
                                         use_pointcloud=pc, use_contrastive=pc)
    shapes = {k: tuple(v.shape) for k, v in mla.state_dict().items()}
    mla.load_state_dict(recipe.make_state_dict(shapes), strict=True)
    mla.freeze_backbones("pretrain")
    mla.train()
    batch, draws = recipe.make_batch(R=R, ragged=ragged)
    kw = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"], images=batch["images"],
              point_cloud=batch["point_cloud"] if pc else None, actions=batch["actions"], proprio=batch["proprio"], action_masks=batch["action_masks"],
              camera_name=batch["camera_name"], gripper_xyz=None, output_hidden_states=True, repeated_diffusion_steps=R, use_diff=True)
    import builtins
    _print = builtins.print
    builtins.print = lambda *a, **k: None
    try:
        if mode == "A":
            up = lambda m, a: tuple(x.float() if torch.is_tensor(x) and x.is_floating_point() else x for x in a)  # noqa: E731
            mla.vlm.proprio_embedder.register_forward_pre_hook(up)
            mla.vlm.x_embedder.register_forward_pre_hook(up)
            with _Draws(draws, 2 * R):
                ld, out = mla(**kw)
        else:
            mla.to(torch.bfloat16)
            kw["images"] = {k: v.to(torch.bfloat16) for k, v in kw["images"].items()}
            for k in ("actions", "proprio") + (("point_cloud",) if pc else ()):
                kw[k] = kw[k].to(torch.bfloat16)
            with _Draws(draws, 2 * R), torch.autocast("cpu", dtype=torch.bfloat16):
                ld, out = mla(**kw)
        ld["total_loss"].float().backward()
    finally:
        builtins.print = _print
    grads = {k: p.grad for k, p in mla.named_parameters() if p.grad is not None}
    return shapes, ld, grads


def main():
    pc = "--pc" in sys.argv
    eq = "--eq" in sys.argv
    R = int(sys.argv[sys.argv.index("--eq") + 1]) if eq else 2
    res = {"R": np.array(R)} if eq else {}
    for mode in ("A", "C"):
        shapes, ld, grads = run(mode, R=R, pc=pc, ragged=not eq)
        f = lambda t: t.detach().float().numpy()  # noqa: E731
        res[f"{mode}_total_loss"] = f(ld["total_loss"])
        res[f"{mode}_gradnorms"] = np.array([float(grads[k].float().norm()) for k in sorted(grads)], dtype=np.float64)
        for k in sorted(grads):                        # round 6: an A and a C sample of EVERY parameter's gradient (recipe.grad_slice)
            res[f"{mode}_gs::{k}"] = f(recipe.grad_slice(grads[k]))
        for k in SLICES + (PC_SLICES if pc else ()):
            g = grads[k]
            res[f"{mode}_grad::{k}"] = f(g.reshape(g.shape[0], -1)[:16, :64])
        for k in VECS + (PC_VECS if pc else ()):
            res[f"{mode}_grad::{k}"] = f(grads[k].reshape(-1)[:256])
        if mode == "A":
            res["grad_names"] = np.array(sorted(grads))
            res["param_names"] = np.array(sorted(shapes))
            res["param_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes)])
    name = "mla_tiny_e2e_pretrain_eq.npz" if eq else "mla_tiny_e2e_pretrain_pc.npz" if pc else "mla_tiny_e2e_pretrain.npz"
    np.savez_compressed(os.path.join(OUT, name), **res)
    vt = [n for n in res["grad_names"] if "vision_tower_" in str(n)]
    print(name, ": A loss", res["A_total_loss"], "C loss", res["C_total_loss"], "| vision-tower params with grad:", vt)


if __name__ == "__main__":
    main()
