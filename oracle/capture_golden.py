"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference, build container only).

    PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden.py

Inputs and weights come from oracle/recipe.py (pure functions of names + seeds), so only outputs are stored.
Modes: A = fp32 (+ pre-hooks undoing the hard bf16 casts' dtype, keeping their rounding),
                   C = model.to(bf16) + autocast(cpu, bf16) with bf16 float inputs (what FSDP mixed precision feeds).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recipe, ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class _Draws:
    """Feeds the reference's torch.randn_like / torch.randint calls from a fixed list, in call order."""

    def __init__(self, draws, Bp):
        self.randn = [draws["noise"]]
        self.randint = [draws["timestep"], draws["fps_start0"], draws["fps_start1"]]
        self._o = (torch.randn_like, torch.randint)

    def __enter__(self):
        torch.randn_like = lambda x, **k: self.randn.pop(0).to(x.dtype)
        torch.randint = lambda *a, **k: self.randint.pop(0)
        return self

    def __exit__(self, *a):
        torch.randn_like, torch.randint = self._o


def run_reference(mode: str, R: int = 2):
    mla = ref_import.build_reference_mla(recipe.TINY_LLAMA | {"vocab_size": recipe.TINY_LLAMA["vocab_size"] + 1}, recipe.TOKEN_SIZE)
    shapes = {k: tuple(v.shape) for k, v in mla.state_dict().items()}
    sd = recipe.make_state_dict(shapes)
    mla.load_state_dict(sd, strict=True)
    mla.freeze_backbones("finetune")
    mla.train()
    batch, draws = recipe.make_batch(R=R)
    kw = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], labels=batch["labels"],
              images=batch["images"], point_cloud=batch["point_cloud"], actions=batch["actions"], proprio=batch["proprio"],
              action_masks=batch["action_masks"], camera_name=batch["camera_name"], gripper_xyz=None, output_hidden_states=True,
              repeated_diffusion_steps=R, use_diff=True)
    import builtins
    _print = builtins.print
    builtins.print = lambda *a, **k: None  # MLA.forward prints the loss dict every call (model_mla.py:233)
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
            for k in ("point_cloud", "actions", "proprio"):
                kw[k] = kw[k].to(torch.bfloat16)
            with _Draws(draws, 2 * R), torch.autocast("cpu", dtype=torch.bfloat16):
                loss_dict, out = mla(**kw)
        loss_dict["total_loss"].float().backward()
    finally:
        builtins.print = _print
    grads = {k: p.grad for k, p in mla.named_parameters() if p.grad is not None}
    return mla, loss_dict, out, grads


def capture_e2e():
    res = {}
    for mode in ("A", "C"):
        mla, ld, out, grads = run_reference(mode)
        f = lambda t: t.detach().float().numpy()  # noqa: E731
        res[f"{mode}_total_loss"] = f(ld["total_loss"])
        res[f"{mode}_contrastive"] = f(ld["img_pc_contrastive_loss"])
        res[f"{mode}_llm_loss"] = f(out.loss)
        res[f"{mode}_logits_slice"] = f(out.logits[:, -8:, :64])
        res[f"{mode}_hidden8_slice"] = f(out.hidden_states[8][:, 250:270, :32])
        res[f"{mode}_last_hidden_slice"] = f(out.hidden_states[-1][:, -8:, :32])
        for k in ("vlm.final_layer.mlp.fc2.weight", "vlm.final_layer.mlp.fc1.bias", "vlm.x_embedder.mlp.fc1.weight",
                  "vlm.llm_backbone.llm.model.norm.weight", "vlm.llm_backbone.llm.model.layers.8.input_layernorm.weight"):
            res[f"{mode}_grad::{k}"] = f(grads[k])
        for k in ("vlm.llm_backbone.llm.model.layers.0.self_attn.q_proj.weight", "vlm.llm_backbone.llm.model.layers.3.mlp.down_proj.weight",
                  "vlm.llm_backbone.llm.model.layers.8.mlp.gate_proj.weight", "vlm.projector_2d.mlp.2.weight",
                  "vlm.projector_3d.projector.0.weight",
                  "vlm.llm_backbone.llm.coordinate_aware_contrastive_loss_module.image_projection_head.2.weight"):
            res[f"{mode}_grad::{k}"] = f(grads[k][:16, :64])
        res[f"{mode}_gradnorms"] = np.array([float(grads[k].float().norm()) for k in sorted(grads)], dtype=np.float64)
        for k in sorted(grads):                        # round 6: an A and a C sample of EVERY parameter's gradient (recipe.grad_slice)
            res[f"{mode}_gs::{k}"] = f(recipe.grad_slice(grads[k]))
        if mode == "A":
            res["grad_names"] = np.array(sorted(grads))
    np.savez_compressed(os.path.join(OUT, "mla_tiny_e2e.npz"), **res)
    print("mla_tiny_e2e.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(res.items())[:6]})
    print("  A total", res["A_total_loss"], "contrastive", res["A_contrastive"], "| C total", res["C_total_loss"])


def capture_components():
    ref_import.setup()
    res = {}
    # --- action tokenizer (vla/action_tokenizer.py) -- bit-exact ids
    from vla.action_tokenizer import ActionTokenizer

    class _T:
        vocab_size = 32000

    at = ActionTokenizer(_T())
    a = np.concatenate([np.array([-1, -0.999, -0.5, 0, 1e-9, 0.5, 0.996, 1, 1.5, -2.0]), np.linspace(-1.2, 1.2, 1001)])
    ids = at.tokenizer.vocab_size - np.digitize(np.clip(a, -1.0, 1.0), at.bins)
    res["at_actions"], res["at_ids"] = a, ids
    res["at_decoded"] = at.decode_token_ids_to_actions(ids)
    # --- diffusion schedule + q_sample
    from models.diffusion import create_diffusion
    d = create_diffusion(timestep_respacing="", noise_schedule="squaredcos_cap_v2", diffusion_steps=100, sigma_small=True, learn_sigma=False)
    res["betas"], res["sqrt_ac"], res["sqrt_1mac"] = d.betas, d.sqrt_alphas_cumprod, d.sqrt_one_minus_alphas_cumprod
    g = torch.Generator().manual_seed(7)
    x0, nz, t = torch.randn(8, 1, 7, generator=g), torch.randn(8, 1, 7, generator=g), torch.randint(0, 100, (8,), generator=g)
    res["qs_x0"], res["qs_noise"], res["qs_t"], res["qs_out"] = x0.numpy(), nz.numpy(), t.numpy(), d.q_sample(x0, t, nz).numpy()
    # --- camera projection for the three cameras
    from models.mla.fuser.camera import get_camera_params, get_projection_func
    pts = torch.rand(4, 256, 3, generator=g) * torch.tensor([1.2, 1.6, 1.0]) + torch.tensor([-0.3, -0.8, 0.5])
    res["proj_pts"] = pts.numpy()
    for cam in ("rlbench_front", "franka_right", "franka_front"):
        p = get_camera_params(cam)
        idx, valid = get_projection_func(cam)(pts, p.K, p.R, p.t, image_size_resize=(672, 672),
                                              vision_strides={"patch_stride": 14, "conv_stride": 3})
        res[f"proj_idx_{cam}"], res[f"proj_valid_{cam}"] = idx.numpy(), valid.numpy()
    # --- point tokenizer (indices + tokens) and vision tokenizer (tokens) on recipe weights
    from models.mla.image.vision_tokenizer import MLP_GELU, VisionTokenizer
    from models.mla.pointcloud.backbone.pointvit import PointTokenizer
    batch, draws = recipe.make_batch(R=1)
    pt = PointTokenizer()
    pt.load_state_dict(recipe.make_state_dict({"vlm.vision_tower_3d." + k: v.shape for k, v in pt.state_dict().items()}
                                              ) and {k: recipe.det_weight("vlm.vision_tower_3d." + k, v.shape) for k, v in pt.state_dict().items()})
    pt.train()
    import models.mla.pointcloud.backbone.Point_PN as PN
    rec = {}
    o_fps, o_knn = PN.furthest_point_sample, PN.knn_point
    def fps_rec(xyz, n):
        r = o_fps(xyz, n); rec.setdefault("fps", []).append(r); return r
    def knn_rec(k, xyz, new):
        r = o_knn(k, xyz, new); rec.setdefault("knn", []).append(r); return r
    PN.furthest_point_sample, PN.knn_point = fps_rec, knn_rec
    starts = [draws["fps_start0"], draws["fps_start1"]]
    o_ri = torch.randint
    torch.randint = lambda *a, **k: starts.pop(0)
    try:
        with torch.no_grad():
            tok, ctr = pt(batch["point_cloud"])
    finally:
        torch.randint = o_ri
        PN.furthest_point_sample, PN.knn_point = o_fps, o_knn
    res["pt_tokens_slice"], res["pt_centers"] = tok[:, :, :64].numpy(), ctr.numpy()
    res["pt_fps0"], res["pt_fps1"] = rec["fps"][0].numpy(), rec["fps"][1].numpy()
    res["pt_knn0_sorted"] = np.sort(rec["knn"][0].numpy(), -1).astype(np.int16)
    res["pt_knn1_sorted"] = np.sort(rec["knn"][1].numpy(), -1).astype(np.int16)
    vt = VisionTokenizer(1024)
    vt.load_state_dict({k: recipe.det_weight("vlm.vision_tower_2d." + k, v.shape) for k, v in vt.state_dict().items()})
    proj = MLP_GELU(1024, recipe.TOKEN_SIZE, 2)
    proj.load_state_dict({k: recipe.det_weight("vlm.projector_2d." + k, v.shape) for k, v in proj.state_dict().items()})
    with torch.no_grad():
        toks, hw = vt(batch["images"]["front_image"], proj)
    res["vt_tokens_slice"] = torch.stack(toks)[:, :, :64].numpy()
    # --- FinalLayer (models/diffusion/models.py:173-189) on NON-zero-mean rows with outlier channels: the input on which timm
    # 0.9.10's torch.var-based RmsNorm and a mean-of-squares RMS norm visibly differ (Llama hidden states look like this)
    from models.diffusion.models import FinalLayer
    fl = FinalLayer(256, 7)
    fl.load_state_dict({k: recipe.det_weight("vlm.final_layer." + k, v.shape) for k, v in fl.state_dict().items()})
    with torch.no_grad():
        fl.norm_final.weight.copy_(1.0 + 0.25 * torch.randn(256, generator=g))
    xf = torch.randn(12, 256, generator=g) * torch.linspace(0.5, 2.0, 12)[:, None] + torch.linspace(-3.0, 3.0, 12)[:, None]
    xf[:, 7] += 20.0
    xf[:, 100] -= 12.0
    xf.requires_grad_(True)
    nf = fl.norm_final(xf)
    yf = fl.mlp(nf)
    gy = torch.randn(yf.shape, generator=g)
    gx, gnw, gfc1 = torch.autograd.grad(yf, [xf, fl.norm_final.weight, fl.mlp.fc1.weight], gy)
    res["fl_x"], res["fl_norm_w"], res["fl_normed"], res["fl_y"], res["fl_gy"] = (xf.detach().numpy(), fl.norm_final.weight.detach().numpy(),
                                                                                   nf.detach().numpy(), yf.detach().numpy(), gy.numpy())
    res["fl_gx"], res["fl_g_norm_w"], res["fl_g_fc1_w"] = gx.numpy(), gnw.numpy(), gfc1.numpy()
    np.savez_compressed(os.path.join(OUT, "components.npz"), **res)
    print("components.npz keys:", len(res))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    capture_components()
    capture_e2e()
