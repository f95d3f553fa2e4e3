#!/usr/bin/env python
"""profiles/rNN_parity_table.txt: the measured end-to-end parity numbers behind the bounds of tests/test_model_gpu.py::
test_mla_e2e_against_reference_golden and tests/test_generation_gpu.py::test_mla_e2e_post_training.

For every compared tensor: err(hip, A) and the yardstick err(C, A), where A = the reference in fp32 and C = the reference in its own
GPU arithmetic (model.to(bf16) + bf16 autocast), both captured from the imported reference (tests/golden/mla_tiny_e2e*.npz).
err = Frobenius-relative for tensors, |x - A| / A for gradient norms. Run on the GPU box:
    python tools/parity_table.py > gpurun_out/parity_table.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
G = os.path.join(ROOT, "tests", "golden")


def err(a, ref):
    return float(np.linalg.norm(a - ref) / (np.linalg.norm(ref) + 1e-30))


def sft(dev):
    import test_model_gpu as T
    e2e = np.load(os.path.join(G, "mla_tiny_e2e.npz"), allow_pickle=True)
    m, ld, out = T._run_hip_e2e(dev)
    print("== tiny MLA SFT step (BASELINE configs[0] shapes) vs reference golden: hip | mode C | ratio hip / C")
    for k, got, a, c in (("total_loss", ld["total_loss"], "A_total_loss", "C_total_loss"),
                         ("img_pc_contrastive_loss", ld["img_pc_contrastive_loss"], "A_contrastive", "C_contrastive"),
                         ("llm_loss", out.loss, "A_llm_loss", "C_llm_loss")):
        A, C = float(e2e[a]), float(e2e[c])
        print(f"loss {k:<28} |hip - A| {abs(float(got) - A):.2e} | |C - A| {abs(C - A):.2e}   (A = {A:.6f})")
    for name, got in (("hidden8_slice", out.hidden_states[8][:, 250:270, :32]), ("last_hidden_slice", out.hidden_states[-1][:, -8:, :32]),
                      ("logits_slice", out.logits[:, -8:, :64])):
        A, C = e2e["A_" + name], e2e["C_" + name]
        g = got.detach().float().cpu().numpy()
        if name != "hidden8_slice":
            keep = np.ones(A.shape[:2], dtype=bool)
            keep[1, -3:] = keep[3, -3:] = False
            A, C, g = A[keep], C[keep], g[keep]
        print(f"act  {name:<28} {err(g, A):.2e} | {err(C, A):.2e} | {err(g, A) / err(C, A):.2f}")
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    for key in e2e.files:
        if key.startswith("A_grad::"):
            n = key[len("A_grad::"):]
            A, C = e2e[key], e2e["C_grad::" + n]
            g = grads[n].float().cpu()
            g = (g if tuple(g.shape) == A.shape else g[:16, :64]).numpy()
            print(f"grad {n:<95} {err(g, A):.2e} | {err(C, A):.2e} | {err(g, A) / err(C, A):.2f}")
    names = [str(n) for n in e2e["grad_names"]]
    gn = np.array([float(grads[k].float().norm()) for k in names])
    A, C = e2e["A_gradnorms"], e2e["C_gradnorms"]
    relA, relC = np.abs(gn - A) / (A + 1e-12), np.abs(C - A) / (A + 1e-12)
    print(f"gradient norms over {len(names)} parameters: median rel hip {np.median(relA):.2e} | mode C {np.median(relC):.2e}; "
          f"max hip {relA.max():.2e} | max C {relC.max():.2e}")
    print("parameters with rel(hip) > 2 x rel(C) (norm A, rel hip, rel C):")
    for n, a, ra, rc in sorted(zip(names, A, relA, relC), key=lambda t: -t[2]):
        if ra > 2 * rc:
            print(f"   {n:<100} |A| {a:.3e}  hip {ra:.2e}  C {rc:.2e}")


def gen(dev):
    import test_generation_gpu as T
    m, gold = T.build_tiny_mla_gen(dev)
    from oracle import recipe
    batch, draws = recipe.make_batch(R=2, with_next=True)
    m.vlm.vision_tower_3d.fps_starts_override = [draws["fps_start0"], draws["fps_start1"]]
    to = lambda v: v.to(dev)  # noqa: E731
    BF = torch.bfloat16
    ld, out = m(input_ids=to(batch["input_ids"]), attention_mask=to(batch["attention_mask"]), labels=to(batch["labels"]),
                images={"front_image": to(batch["images"]["front_image"]).to(BF)}, next_images=to(batch["next_images"]).to(BF),
                point_cloud=to(batch["point_cloud"]), next_point_cloud=to(batch["next_point_cloud"]), actions=to(batch["actions"]),
                proprio=to(batch["proprio"]), action_masks=to(batch["action_masks"]), camera_name=batch["camera_name"],
                repeated_diffusion_steps=2, use_diff=True, noise=to(draws["noise"]), timestep=to(draws["timestep"]))
    ld["total_loss"].backward()
    print("\n== tiny MLA post-training step (configs[3] scaled down) vs reference golden: hip | mode C")
    for k in ("total_loss", "image_gen_loss", "point_cloud_gen_loss", "img_pc_contrastive_loss"):
        A, C = float(gold["A_" + k]), float(gold["C_" + k])
        print(f"loss {k:<28} |hip - A| {abs(float(ld[k]) - A):.2e} | |C - A| {abs(C - A):.2e}   (A = {A:.6f})")
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    names = [str(n) for n in gold["grad_names"]]
    A, C = gold["A_gradnorms"], gold["C_gradnorms"]
    gn = np.array([float(grads[k].float().norm()) if k in grads else 0.0 for k in names])
    live = A > 0
    relA, relC = np.abs(gn - A)[live] / A[live], np.abs(C - A)[live] / A[live]
    ln = np.array(names)[live]
    print(f"gradient norms over {live.sum()} live parameters: median rel hip {np.median(relA):.2e} | mode C {np.median(relC):.2e}; "
          f"max hip {relA.max():.2e} | max C {relC.max():.2e}")
    print("parameters with rel(hip) > 2 x rel(C) (norm A, rel hip, rel C):")
    for n, a, ra, rc in sorted(zip(ln, A[live], relA, relC), key=lambda t: -t[2]):
        if ra > 2 * rc:
            print(f"   {n:<100} |A| {a:.3e}  hip {ra:.2e}  C {rc:.2e}")
    print("largest rel(C) (the yardstick's own tail):")
    for n, a, ra, rc in sorted(zip(ln, A[live], relA, relC), key=lambda t: -t[3])[:8]:
        print(f"   {n:<100} |A| {a:.3e}  hip {ra:.2e}  C {rc:.2e}")


def heads(dev):
    """tests/test_generation_gpu.py::test_generation_heads_against_reference_golden: the generation manager alone vs generation.npz"""
    import test_generation_gpu as T
    from mla_amd import ops
    from oracle import recipe
    BF = torch.bfloat16
    gold = np.load(os.path.join(G, "generation.npz"), allow_pickle=True)
    pfx = "vlm.generation_manager."
    mgr = T._build_manager(dev)
    mgr.load_state_dict({k: recipe.det_weight(pfx + k, v.shape) for k, v in mgr.state_dict().items()}, strict=True)
    T._zero_dropout(mgr)
    mgr.train().to(dev)
    for p in mgr.parameters():
        p.data = p.data.to(BF)
    hidden, curr, nxt, npc = T._gen_inputs()
    hd = hidden.to(dev, BF).requires_grad_()
    outs = mgr(llm_hidden_states=hd)
    loss_img, parts = ops.ImageGenLossFn.apply(outs["delta_raw"], curr.to(dev, BF), nxt.to(dev, BF), 42, 5.0)
    loss_pc = ops.ChamferFn.apply(outs["pointcloud_coord_generation"], npc.to(dev))
    (loss_img + loss_pc).backward()
    print("\n== generation heads alone (tiny dims) vs reference golden generation.npz: hip | mode C | ratio")
    for key, got in (("image_gen_loss", loss_img), ("point_cloud_gen_loss", loss_pc)):
        A, C = float(gold["A_" + key]), float(gold["C_" + key])
        print(f"loss {key:<28} |hip - A| {abs(float(got) - A):.2e} | |C - A| {abs(C - A):.2e}")
    delta = (torch.tanh(outs["delta_raw"][..., :5292].float()) * 5.0)[:, ::16, ::97].detach().cpu().numpy()
    pts = outs["pointcloud_coord_generation"].detach().float().cpu().numpy()
    hg = hd.grad.float().cpu().numpy()
    for name, got, a, c in (("delta_slice", delta, "A_delta_slice", "C_delta_slice"), ("points", pts, "A_points", "C_points"),
                            ("hidden_grad", hg, "A_hidden_grad", "C_hidden_grad")):
        print(f"act  {name:<28} {err(got, gold[a]):.2e} | {err(gold[c], gold[a]):.2e} | {err(got, gold[a]) / err(gold[c], gold[a]):.2f}")
    grads = {k: p.grad for k, p in mgr.named_parameters() if p.grad is not None}
    names = [str(n) for n in gold["grad_names"]]
    A, C = gold["A_gradnorms"], gold["C_gradnorms"]
    gn = np.array([float(grads[k].float().norm()) if k in grads else 0.0 for k in names])
    live = A > 1e-6
    relA, relC = np.abs(gn - A)[live] / A[live], np.abs(C - A)[live] / A[live]
    q90 = float(np.quantile(relC, 0.9))
    print(f"gradient norms over {live.sum()} live parameters: median rel hip {np.median(relA):.2e} | mode C {np.median(relC):.2e}; q90(C) {q90:.2e}; "
          f"violations of rel(hip) <= 2 max(rel C, q90): {[(n, float(a), float(c)) for n, a, c in zip(np.array(names)[live], relA, relC) if a > 2 * max(c, q90)]}")
    print("not live (|A| <= 1e-6):", [(n, float(a), float(c), float(h)) for n, a, c, h in zip(names, A, C, gn) if a <= 1e-6])
    for key in gold.files:
        if key.startswith("A_grad::"):
            n = key[len("A_grad::"):]
            ref = gold[key]
            g = grads[n].float().cpu()
            got = (g.reshape(g.shape[0], -1)[:16, :64] if ref.ndim == 2 else g.reshape(-1)[:256]).numpy()
            c = gold["C_grad::" + n] if ("C_grad::" + n) in gold.files else None
            print(f"grad {n:<80} {err(got, ref):.2e} | " + (f"{err(c, ref):.2e} | {err(got, ref) / err(c, ref):.2f}" if c is not None else "no mode-C slice in the golden"))
    bn = mgr.pointcloud_gen_module.future_predictor[1]
    print("bn running_mean", err(bn.running_mean.float().cpu().numpy(), gold["A_bn_running_mean"]), "running_var", err(bn.running_var.float().cpu().numpy(), gold["A_bn_running_var"]))


def strict(dev):
    """Round 6: err(hip, A) <= 2 x err(C, A) on a gradient sample of EVERY parameter of all five end-to-end
    goldens (tests/parity_util.py), not on 11 captured slices."""
    import parity_util as P
    import test_generation_gpu as TG
    import test_model_gpu as TM
    import test_pretrain_gpu as TP
    import test_tactile_gpu as TT
    print("\n\n#### strict per-parameter gradient parity (every parameter, recipe.grad_slice samples) ####")
    e2e = np.load(os.path.join(G, "mla_tiny_e2e.npz"), allow_pickle=True)
    m, ld, out = TM._run_hip_e2e(dev)
    print(P.format_rows(P.grad_sample_rows({k: p.grad for k, p in m.named_parameters() if p.grad is not None}, e2e),
                        "tiny MLA SFT step (configs[0] shapes; mla_tiny_e2e.npz)"))
    del m, ld, out
    m, ld, gold, _, _ = TG.run_post_training_e2e(dev)
    print(P.format_rows(P.grad_sample_rows({k: p.grad for k, p in m.named_parameters() if p.grad is not None}, gold),
                        "tiny MLA post-training step (configs[3] scaled down; mla_tiny_e2e_gen.npz)"))
    del m, ld
    for pc in (False, True):
        m, ld, gold = TP.run_pretrain_e2e(dev, pc)
        print(P.format_rows(P.grad_sample_rows({k: p.grad for k, p in m.named_parameters() if p.grad is not None}, gold),
                            f"tiny MLA stage pretrain, use_pointcloud={pc} (mla_tiny_e2e_pretrain{'_pc' if pc else ''}.npz)"))
        del m, ld
    m, ld, gold = TT.run_tactile_e2e(dev)
    print(P.format_rows(P.grad_sample_rows({k: p.grad for k, p in m.named_parameters() if p.grad is not None}, gold),
                        "tiny MLA with tactile + generation heads (mla_tiny_e2e_tactile.npz)"))
    # the one tensor over 2 x mode C: pad-row semantics, not arithmetic (tests/test_tactile_gpu.py::test_mla_e2e_tactile)
    from oracle import mla_oracle, recipe
    KEY = "vlm.generation_manager.tactile_gen_module.decoder.layers.0.multihead_attn.in_proj_weight"
    batch, draws = TT.run_tactile_e2e.last_inputs
    mine = recipe.grad_slice(dict(m.named_parameters())[KEY].grad.float().cpu()).numpy()
    A, C = gold["A_gs::" + KEY], gold["C_gs::" + KEY]
    print(f"  named exception {KEY}:")
    for zp, tag in ((False, "eager pad rows (the golden's semantics)"), (True, "flash / varlen pad rows (the kernels' and the reference GPU path's)")):
        sd = {k: recipe.det_weight(k, v.shape) for k, v in m.state_dict().items()}
        sd[KEY].requires_grad_(True)
        ref = mla_oracle.mla_forward(sd, batch, draws, 9, 2, 1e-5, 2, zero_pad_rows=zp, use_tactile=True, gen_tactile=True)
        g = recipe.grad_slice(torch.autograd.grad(ref["total_loss"], sd[KEY])[0]).numpy()
        print(f"    fp32 oracle, {tag:<72} err(oracle, A) {err(g, A):.2e}   err(hip, oracle) {err(mine, g):.2e}   yardstick err(C, A) {err(C, A):.2e}")


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    sft(dev)
    gen(dev)
    heads(dev)
    strict(dev)
