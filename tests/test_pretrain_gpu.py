"""GPU parity of stage "pretrain" (trainable vision tokenizer; BASELINE configs[4] shape): the two backward kernels vs autograd of
the oracle, the tokenizer's parameter gradients vs the oracle, and the whole tiny-MLA step vs the reference golden
(tests/golden/mla_tiny_e2e_pretrain.npz) with the C-vs-A yardstick."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import fro_rel
from oracle import recipe
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
BF = torch.bfloat16


def test_local_attention_and_pool_backward(dev):
    from mla_amd import ops
    B, gh, gw, cs, C, heads = 2, 6, 6, 3, 256, 8
    g = torch.Generator().manual_seed(0)
    q = (torch.randn(B * 4, C, generator=g)).to(BF)
    kv = (torch.randn(B * gh * gw, 2 * C, generator=g)).to(BF)
    do = (torch.randn(B * 4, C, generator=g)).to(BF)
    scale = C ** -0.5 * 8          # sharper softmax than the model's, to exercise dS

    def ref(qr, kvr):
        qh = qr.view(B, 2, 2, heads, C // heads)
        kvh = kvr.view(B, 2, cs, 2, cs, 2, heads, C // heads).permute(0, 1, 3, 2, 4, 5, 6, 7).reshape(B, 2, 2, cs * cs, 2, heads, C // heads)
        s = torch.einsum("bijhd,bijnhd->bijhn", qh * scale, kvh[:, :, :, :, 0])
        p = torch.softmax(s, -1)
        return torch.einsum("bijhn,bijnhd->bijhd", p, kvh[:, :, :, :, 1]).reshape(B * 4, C)
    qr, kvr = q.float().requires_grad_(), kv.float().requires_grad_()
    out_r = ref(qr, kvr)
    out_r.backward(do.float())
    qd, kvd = q.to(dev).requires_grad_(), kv.to(dev).requires_grad_()
    out = ops.LocalAttnFn.apply(qd, kvd, B, gh, gw, cs, heads, scale)
    out.backward(do.to(dev))
    assert fro_rel(out, out_r) < 5e-3
    assert fro_rel(qd.grad, qr.grad) < 1e-2 and fro_rel(kvd.grad, kvr.grad) < 1e-2
    x = torch.randn(B * gh * gw, C, generator=g).to(BF)
    xr = x.float().requires_grad_()
    yr = F.avg_pool2d(xr.view(B, gh, gw, C).permute(0, 3, 1, 2), cs, cs).permute(0, 2, 3, 1).reshape(B * 4, C)
    yr.backward(do.float())
    xd = x.to(dev).requires_grad_()
    y = ops.AvgPoolTokensFn.apply(xd, B, gh, gw, cs)
    y.backward(do.to(dev))
    assert fro_rel(y, yr) < 4e-3 and fro_rel(xd.grad, xr.grad) < 4e-3


def test_trainable_vision_tokenizer_gradients_vs_oracle(dev):
    from mla_amd.vision_tokenizer import MLP_GELU, VisionTokenizer
    vt, proj = VisionTokenizer(1024), MLP_GELU(1024, recipe.TOKEN_SIZE, 2)
    sd_v = {k: recipe.det_weight("vlm.vision_tower_2d." + k, v.shape) for k, v in vt.state_dict().items()}
    sd_p = {k: recipe.det_weight("vlm.projector_2d." + k, v.shape) for k, v in proj.state_dict().items()}
    vt.load_state_dict(sd_v); proj.load_state_dict(sd_p)
    vt.to(dev).to(BF); proj.to(dev).to(BF)
    batch, _ = recipe.make_batch(R=1)
    img = batch["images"]["front_image"]
    toks, _ = vt(img.to(dev), proj)
    out = torch.stack(toks)
    gout = recipe.det_randn("pretrain.gout", tuple(out.shape)).to(BF)
    out.backward(gout.to(dev))
    # oracle: same function, fp32 autograd
    w = {"patch_w": sd_v["patch_embedding.weight"].clone().requires_grad_(),
         "q_ln_w": sd_v["local_attention.q.0.weight"].clone().requires_grad_(), "q_ln_b": sd_v["local_attention.q.0.bias"].clone().requires_grad_(),
         "q_w": sd_v["local_attention.q.1.weight"].clone().requires_grad_(),
         "kv_ln_w": sd_v["local_attention.kv.0.weight"].clone().requires_grad_(), "kv_ln_b": sd_v["local_attention.kv.0.bias"].clone().requires_grad_(),
         "kv_w": sd_v["local_attention.kv.1.weight"].clone().requires_grad_(),
         "proj_w": sd_v["local_attention.proj.weight"].clone().requires_grad_(), "proj_b": sd_v["local_attention.proj.bias"].clone().requires_grad_()}
    pj = dict(w0=sd_p["mlp.0.weight"], b0=sd_p["mlp.0.bias"], w2=sd_p["mlp.2.weight"], b2=sd_p["mlp.2.bias"])
    ref = O.vision_tokenizer(img, w, pj)
    ref.backward(gout.float())
    assert fro_rel(out, ref.detach()) < 2e-2
    pairs = {"patch_embedding.weight": "patch_w", "local_attention.q.0.weight": "q_ln_w", "local_attention.q.0.bias": "q_ln_b",
             "local_attention.q.1.weight": "q_w", "local_attention.kv.0.weight": "kv_ln_w", "local_attention.kv.0.bias": "kv_ln_b",
             "local_attention.kv.1.weight": "kv_w", "local_attention.proj.weight": "proj_w", "local_attention.proj.bias": "proj_b"}
    got = dict(vt.named_parameters())
    for name, key in pairs.items():
        assert got[name].grad is not None, name
        assert fro_rel(got[name].grad, w[key].grad) < 4e-2, (name, fro_rel(got[name].grad, w[key].grad))
    for name in ("class_embedding", "split_embedding", "global_attention.proj.weight"):
        assert got[name].grad is None          # unused by the forward in the reference as well (vision_tokenizer.py:142,149)


def test_point_gather_and_maxpool_backward(dev):
    """lga_prep / maxpool_k backward kernels (Point_PN.py:115-158, :166-169) against autograd of the same index arithmetic."""
    from mla_amd import hip, ops
    B, N, G, K, C = 2, 64, 32, 9, 24
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(B, N, 3, generator=g)
    feats = torch.randn(B, N, C, generator=g).to(BF)
    fps = torch.stack([torch.randperm(N, generator=g)[:G] for _ in range(B)])
    knn = torch.randint(0, N, (B, G, K), generator=g, dtype=torch.int32)
    fd = feats.to(dev).requires_grad_()
    rows, _ = ops.LgaPrepFn.apply(xyz.to(dev), fd, fps.to(dev), knn.to(dev), 1000.0, 100.0)
    do = torch.randn(rows.shape, generator=g).to(BF)
    rows.backward(do.to(dev))
    fr = feats.float().requires_grad_()
    bi = torch.arange(B)[:, None, None]
    nb = fr[bi, knn.long()]                                      # [B, G, K, C]
    ct = fr[torch.arange(B)[:, None], fps][:, :, None].expand(-1, -1, K, -1)
    (torch.cat([nb, ct], -1).reshape(B * G * K, 2 * C) * do.float()).sum().backward()
    assert fro_rel(fd.grad, fr.grad) < 4e-3
    # round 4: a gather with every sum in a fixed order (no fp32 atomics) -- repeated calls are bit-equal, also with a point that is
    # the neighbour of every group (the longest list) and with indices repeated inside a group (torch.randint above draws some)
    knn_hot = knn.clone()
    knn_hot[:, :, 0] = 5
    for kk in (knn, knn_hot):
        a = hip.lga_prep_bwd(do.to(dev), fps.to(dev), kk.to(dev), B, N, C)
        for _ in range(3):
            assert torch.equal(a, hip.lga_prep_bwd(do.to(dev), fps.to(dev), kk.to(dev), B, N, C))
        ref32 = torch.zeros(B, N, C)
        dof = do.float().view(B, G, K, 2 * C)
        for b_ in range(B):
            ref32[b_].index_add_(0, kk[b_].reshape(-1).long(), dof[b_, :, :, :C].reshape(G * K, C))
            ref32[b_].index_add_(0, fps[b_], dof[b_, :, :, C:].sum(1))
        assert fro_rel(a, ref32) < 1e-6
    x = torch.randn(B * G * K, 2 * C, generator=g).to(BF)
    x[K:2 * K, 0] = x[K, 0]                                       # a tie: the first maximum takes the gradient
    xd = x.to(dev).requires_grad_()
    y = ops.MaxPoolKFn.apply(xd, B * G, K)
    dy = torch.randn(y.shape, generator=g).to(BF)
    y.backward(dy.to(dev))
    xr = x.float()
    am = xr.view(B * G, K, 2 * C).argmax(1)                       # first occurrence
    ref = torch.zeros(B * G, K, 2 * C).scatter_(1, am[:, None], dy.float()[:, None]).view(B * G * K, 2 * C)
    assert torch.equal(xd.grad.float().cpu(), ref)


def test_trainable_point_tokenizer_gradients_vs_oracle(dev):
    from mla_amd.point_tokenizer import PointTokenizer
    from oracle.mla_oracle import point_weights
    pt = PointTokenizer()
    sd = {k: recipe.det_weight("vlm.vision_tower_3d." + k, v.shape) for k, v in pt.state_dict().items()}
    pt.load_state_dict(sd)
    pt.to(dev)
    for p in pt.parameters():
        p.data = p.data.to(BF)
    batch, draws = recipe.make_batch(R=1)
    pc = batch["point_cloud"]
    starts = [draws["fps_start0"], draws["fps_start1"]]
    pt.fps_starts_override = starts
    tokens, centres = pt(pc.to(dev))
    gout = recipe.det_randn("pretrain.pc.gout", tuple(tokens.shape)).to(BF)
    tokens.backward(gout.to(dev))
    ref_sd = {"vlm.vision_tower_3d." + k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    w = point_weights(ref_sd)
    rt, rc, dbg = O.point_tokenizer(pc, w, starts)
    rt.backward(gout.float())
    for (f, k), (rf, rk) in zip(pt.last_indices, dbg):
        assert torch.equal(f.cpu(), rf)
        # neighbour SETS (the order inside a group is irrelevant under the max-pool; a tie at the 81st neighbour may swap one index)
        assert (torch.sort(k.cpu().long(), -1)[0] == torch.sort(rk.long(), -1)[0]).all(-1).float().mean() > 0.999
    assert fro_rel(tokens, rt.detach()) < 2e-2 and torch.allclose(centres.cpu(), rc, atol=1e-6)
    # yardstick: the same oracle under bf16 autocast. The max over the 81 neighbours picks another neighbour when bf16 rounding
    # reorders near-ties, which moves whole gradient rows: the reference's own bf16 run is 0.57-0.88 (relative, element-wise) away
    # from its fp32 run on these weights (tests/golden/mla_tiny_e2e_pretrain_pc.npz, C vs A), so the bound is relative to that.
    c_sd = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in ref_sd.items()}
    with torch.autocast("cpu", dtype=BF):
        ct, _, _ = O.point_tokenizer(pc, point_weights(c_sd), starts)
    ct.float().backward(gout.float())
    got = dict(pt.named_parameters())
    report = {}
    for name, p in got.items():
        r = ref_sd["vlm.vision_tower_3d." + name].grad
        if name in ("cls_token", "pos_embed", "norm.weight", "norm.bias"):
            assert p.grad is None and r is None          # unused by the forward (pointvit.py:59-82)
            continue
        assert p.grad is not None, name
        if name.endswith(".0.bias"):                     # conv bias in front of a train-mode BatchNorm: exactly-zero gradient
            assert float(p.grad.float().norm()) < 1e-2 * float(got[name.replace(".0.bias", ".0.weight")].grad.float().norm()), name
            continue
        mine, yard = fro_rel(p.grad.reshape(r.shape), r), fro_rel(c_sd["vlm.vision_tower_3d." + name].grad, r)
        report[name] = (mine, yard)
        assert mine < 1.5 * yard + 2e-2, (name, mine, yard)
        assert abs(float(p.grad.float().norm()) / float(r.norm()) - 1) < 0.1, name
    assert report["proj.weight"][0] < 1e-2               # no max-pool below it: tight


def run_pretrain_e2e(dev, pc, eq=False, share_prefix=False):
    """Builds the stage-"pretrain" tiny MLA, runs forward + backward on the recipe batch; returns (model, loss_dict, golden).
    eq: the unpadded batch with R = 4 diffusion repeats (mla_tiny_e2e_pretrain_eq.npz); share_prefix: the opt-in shared-prefix forward."""
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    gold = np.load(os.path.join(G, "mla_tiny_e2e_pretrain_eq.npz" if eq else "mla_tiny_e2e_pretrain_pc.npz" if pc else "mla_tiny_e2e_pretrain.npz"),
                   allow_pickle=True)
    R = int(gold["R"]) if eq else 2
    bb = LLaMa2LLMBackbone(config=LlamaConfig(**recipe.TINY_LLAMA, activation_save_level=2), pad_to_multiple_of=1)
    vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=pc, use_contrastive=pc, use_generation=False)
    m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=pc, use_contrastive=pc)
    mine = {k: str(tuple(v.shape)) for k, v in m.state_dict().items()}
    assert mine == {str(n): str(s) for n, s in zip(gold["param_names"], gold["param_shapes"])}
    m.load_state_dict({k: recipe.det_weight(k, v.shape) for k, v in m.state_dict().items()}, strict=True)
    m.freeze_backbones("pretrain")
    m.train().to(dev)
    for p in m.parameters():
        p.data = p.data.to(BF)
    m.share_prefix = share_prefix
    batch, draws = recipe.make_batch(R=R, ragged=not eq)
    to = lambda v: v.to(dev)  # noqa: E731
    if pc:
        m.vlm.vision_tower_3d.fps_starts_override = [draws["fps_start0"], draws["fps_start1"]]
    ld, out = m(input_ids=to(batch["input_ids"]), attention_mask=to(batch["attention_mask"]), labels=to(batch["labels"]),
                images={"front_image": to(batch["images"]["front_image"])}, point_cloud=to(batch["point_cloud"]) if pc else None,
                actions=to(batch["actions"]), proprio=to(batch["proprio"]),
                action_masks=to(batch["action_masks"]), camera_name=batch["camera_name"], repeated_diffusion_steps=R, use_diff=True,
                noise=to(draws["noise"]), timestep=to(draws["timestep"]))
    ld["total_loss"].backward()
    run_pretrain_e2e.last_output = out
    return m, ld, gold


@pytest.mark.parametrize("pc", [False, True])
def test_mla_e2e_pretrain_stage(dev, pc):
    m, ld, gold = run_pretrain_e2e(dev, pc)
    tolL = 2 * abs(float(gold["C_total_loss"]) - float(gold["A_total_loss"])) + 2e-2
    assert abs(float(ld["total_loss"]) - float(gold["A_total_loss"])) < tolL
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    names = [str(n) for n in gold["grad_names"]]
    assert sorted(grads) == names, sorted(set(names) ^ set(grads))
    gn = np.array([float(grads[k].float().norm()) for k in names])
    # a conv bias in front of a train-mode BatchNorm has an exactly-zero gradient (the reference holds 1e-7 rounding noise there)
    dead = np.array([n.endswith(".0.bias") and "EncP" in n for n in names])
    assert (gn[dead] < 1e-3).all()
    gn[dead] = gold["A_gradnorms"][dead]
    relA = np.abs(gn - gold["A_gradnorms"]) / (gold["A_gradnorms"] + 1e-12)
    relC = np.abs(gold["C_gradnorms"] - gold["A_gradnorms"]) / (gold["A_gradnorms"] + 1e-12)
    assert np.median(relA) < 2 * np.median(relC) + 5e-3, (np.median(relA), np.median(relC))
    assert (relA < 2 * relC + 5e-2).mean() > 0.97, [(n, a, c) for n, a, c in zip(names, relA, relC) if a >= 2 * c + 5e-2][:6]

    def err(a, ref):
        return float(np.linalg.norm(a - ref) / np.linalg.norm(ref))
    for key in gold.files:
        if key.startswith("A_grad::"):
            n = key[len("A_grad::"):]
            if n.endswith(".0.bias") and "EncP" in n:
                continue
            A, C = gold[key], gold["C_grad::" + n]
            g = grads[n].float().cpu()
            got = (g.reshape(g.shape[0], -1)[:16, :64] if A.ndim == 2 else g.reshape(-1)[:256]).numpy()
            assert err(got, A) < 2 * err(C, A) + 3e-2, (n, err(got, A), err(C, A))
    # Round 6: the strict per-tensor yardstick, no floor, on a gradient sample of EVERY parameter (tests/parity_util.py). Measured:
    # 113 / 113 (use_pointcloud=False, median ratio 0.86) and 154 / 154 (trainable point tower, median 0.44) within 2 x mode C.
    from parity_util import grad_sample_rows, strict_violations
    rows = grad_sample_rows(grads, gold)
    assert len(rows) == len(names)
    assert not strict_violations(rows), strict_violations(rows)


def test_shared_prefix_forward_matches_the_reference_golden_like_the_tiled_forward(dev):
    """Round 6, opt-in (`mla.share_prefix = True`): stage "pretrain" without a point cloud (BASELINE configs[4], scripts/pretrain.sh) --
    the R = 4 diffusion copies of a sample differ only in their last rows [t, x, </s>], so ONE sequence [prefix | 4 suffix groups] per
    sample replaces the reference's 4 tiled sequences (models/mla/model_mla.py:148-180). Against the reference golden of the same
    unpadded batch (mla_tiny_e2e_pretrain_eq.npz, captured from the real reference's TILED forward): loss and the strict per-tensor
    gradient yardstick on every parameter, for the shared-prefix forward AND for the tiled forward; the two agree with each other far
    inside that bound; the shared layout executes P + R s rows per sample instead of R (P + s)."""
    from parity_util import grad_sample_rows, strict_violations
    res = {}
    for share in (False, True):
        m, ld, gold = run_pretrain_e2e(dev, False, eq=True, share_prefix=share)
        A, C = float(gold["A_total_loss"]), float(gold["C_total_loss"])
        half_ulp = 2.0 ** (np.floor(np.log2(abs(A))) - 7) / 2                      # mode C's loss is a bf16 number
        assert abs(float(ld["total_loss"]) - A) <= 2 * max(abs(C - A), half_ulp), (share, float(ld["total_loss"]), A, C)
        grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
        assert sorted(grads) == [str(n) for n in gold["grad_names"]]
        rows = grad_sample_rows(grads, gold)
        assert not strict_violations(rows), (share, strict_violations(rows))
        res[share] = (float(ld["total_loss"]), {k: v.float().cpu() for k, v in grads.items()}, np.median([r["ratio"] for r in rows]))
        if share:
            lay = run_pretrain_e2e.last_output.shared_prefix_layout
            assert lay["repeats"] == 4 and lay["executed_rows_per_sample"] == lay["prefix_rows"] + (4 + lay["dummy_groups"]) * lay["suffix_rows"]
            assert lay["executed_rows_per_sample"] % 4 == 0 or lay["dummy_groups"] == 0
            assert lay["tiled_rows_per_sample"] == 4 * (lay["prefix_rows"] + lay["suffix_rows"])
            # the (lazy) language-model loss of the shared layout: the same R * B next-token terms as the tiled layout
            res["llm_loss_shared"] = float(run_pretrain_e2e.last_output.loss)
        else:
            res["llm_loss_tiled"] = float(run_pretrain_e2e.last_output.loss)
    assert abs(res[True][0] - res[False][0]) <= 2e-3 * abs(res[False][0]), (res[True][0], res[False][0])
    assert abs(res["llm_loss_shared"] - res["llm_loss_tiled"]) <= 5e-3 * abs(res["llm_loss_tiled"]), (res["llm_loss_shared"], res["llm_loss_tiled"])
    num = sum(float(((res[True][1][k] - res[False][1][k]) ** 2).sum()) for k in res[True][1]) ** 0.5
    den = sum(float((res[False][1][k] ** 2).sum()) for k in res[True][1]) ** 0.5
    print(f"shared-prefix vs tiled forward: loss {res[True][0]:.6f} vs {res[False][0]:.6f}, all gradients rel {num / den:.2e}; "
          f"median gradient-sample ratio vs mode C: shared {res[True][2]:.2f}, tiled {res[False][2]:.2f}")
    assert num / den < 2e-2


def test_shared_prefix_forward_on_a_ragged_batch_matches_the_reference_golden(dev):
    """The shared-prefix forward on a right-padded batch of different prompt lengths (per-sample prefix lengths, first suffix rows and
    RoPE positions; the reference's pad rows behind </s> are not executed) against the golden the reference's TILED forward produced
    on the same ragged batch (mla_tiny_e2e_pretrain.npz, R = 2): loss and the strict per-tensor gradient yardstick on all 113 parameters."""
    from parity_util import grad_sample_rows, strict_violations
    m, ld, gold = run_pretrain_e2e(dev, False, eq=False, share_prefix=True)
    lay = run_pretrain_e2e.last_output.shared_prefix_layout
    assert isinstance(lay["prefix_rows"], list) and len(set(lay["prefix_rows"])) > 1, lay       # the ragged path really ran
    A, C = float(gold["A_total_loss"]), float(gold["C_total_loss"])
    half_ulp = 2.0 ** (np.floor(np.log2(abs(A))) - 7) / 2
    assert abs(float(ld["total_loss"]) - A) <= 2 * max(abs(C - A), half_ulp), (float(ld["total_loss"]), A, C)
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    assert sorted(grads) == [str(n) for n in gold["grad_names"]]
    rows = grad_sample_rows(grads, gold)
    assert len(rows) == 113 and not strict_violations(rows), strict_violations(rows)
    print(f"ragged shared prefix: loss {float(ld['total_loss']):.6f} vs A {A:.6f}; median gradient-sample ratio vs mode C {np.median([r['ratio'] for r in rows]):.2f}; layout {lay}")


def test_shared_prefix_steps_through_fsdp_match_tiled_steps(dev):
    """The opt-in shared-prefix forward through FSDPStrategy (fp32 main_grad delivery, fused AdamW, clip; token rows padded to 64 inside
    the decoder layers): two optimizer steps from the same weights, noise and timesteps -- losses, the clipping norm and the updated
    fp32 masters agree with the tiled forward's (masters: AdamW's first steps are +-lr per element whatever the gradient's size, so the
    elements whose gradient sits at bf16 noise level move by up to 2 lr between the two forms: measured 6.9e-4 of the weights' norm at
    lr = 1e-3, bound 2e-3; losses and norms 2e-4 / 5e-4)."""
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    from mla_amd.strategy import FSDPStrategy
    R = 4
    res = {}
    for share in (False, True):
        bb = LLaMa2LLMBackbone(config=LlamaConfig(**recipe.TINY_LLAMA, activation_save_level=0), pad_to_multiple_of=1)     # every layer checkpointed
        vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=False, use_contrastive=False, use_generation=False)
        m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=False, use_contrastive=False)
        m.load_state_dict({k: recipe.det_weight(k, v.shape) for k, v in m.state_dict().items()}, strict=True)
        m.freeze_backbones("pretrain")
        m.share_prefix = share
        strat = FSDPStrategy(m, dev.index or 0, stage="pretrain", global_batch_size=2, per_device_batch_size=2, learning_rate=1e-3, weight_decay=0.01,
                             max_grad_norm=1.0, lr_scheduler_type="constant", enable_gradient_checkpointing=False, repeated_diffusion_steps=R)
        strat.run_setup(100)
        batch, draws = recipe.make_batch(R=R, ragged=False)
        b = {k: v for k, v in batch.items() if k != "point_cloud"}
        orig = m.forward
        m.forward = lambda **kw: orig(**kw, noise=draws["noise"].to(dev), timestep=draws["timestep"].to(dev))
        losses, norms = [], []
        for _ in range(2):
            out = strat.train_step(b)
            losses.append(float(out["total_loss"]))
            norms.append(float(strat.sharded._norm))
        strat.synchronize()
        res[share] = (losses, norms, {u.name: u.master_train.detach().float().cpu() for u in strat.sharded.units if u.trainable})
    for a, b_ in zip(res[True][0] + res[True][1], res[False][0] + res[False][1]):
        assert abs(a - b_) <= 3e-3 * abs(b_), (res[True][:2], res[False][:2])
    num = sum(float(((res[True][2][k] - res[False][2][k]) ** 2).sum()) for k in res[True][2]) ** 0.5
    den = sum(float((res[False][2][k] ** 2).sum()) for k in res[True][2]) ** 0.5
    print(f"shared-prefix vs tiled through FSDPStrategy: losses {res[True][0]} vs {res[False][0]}, norms {res[True][1]} vs {res[False][1]}, masters rel {num / den:.2e}")
    assert num / den < 2e-3


def test_pretrain_step_with_point_tower_through_fsdp(dev):
    """Two optimizer steps of stage "pretrain" with use_pointcloud through FSDPStrategy: conv weights used as matrix views and the
    BatchNorm affine parameters deliver into main_grad, the point tower's weights move, the loss stays finite."""
    import math
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    from mla_amd.strategy import FSDPStrategy
    bb = LLaMa2LLMBackbone(config=LlamaConfig(**recipe.TINY_LLAMA, activation_save_level=1), pad_to_multiple_of=1)
    vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=True, use_contrastive=True, use_generation=False)
    m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=True, use_contrastive=True)
    m.load_state_dict({k: recipe.det_weight(k, v.shape) for k, v in m.state_dict().items()}, strict=True)
    m.freeze_backbones("pretrain")
    strat = FSDPStrategy(vlm=m, device_id=0, stage="pretrain", epochs=1, max_steps=10, global_batch_size=2, per_device_batch_size=2,
                         learning_rate=1e-3, weight_decay=0.0, max_grad_norm=1.0, lr_scheduler_type="constant", warmup_ratio=0.0,
                         repeated_diffusion_steps=2)
    strat.run_setup(n_train_examples=20)
    batch, _ = recipe.make_batch(R=2)
    b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    b["images"] = {"front_image": batch["images"]["front_image"].to(dev)}
    enc = m.vlm.vision_tower_3d.patch_embed.EncP
    conv, bn = enc.LGA_list[0].linear2[0].net1[0], enc.LGA_list[0].linear2[0].net1[1]
    w0, g0 = conv.weight.detach().float().clone(), bn.weight.detach().float().clone()
    l1 = strat.train_step(b)
    assert float(conv.weight.main_grad.abs().max()) > 0 and float(bn.weight.main_grad.abs().max()) > 0
    assert float(enc.raw_point_embed.net[0].weight.main_grad.abs().max()) > 0
    strat.synchronize()                      # the AdamW update runs on the side stream; direct parameter reads wait for it
    assert not torch.equal(conv.weight.detach().float(), w0) and not torch.equal(bn.weight.detach().float(), g0)
    assert int(bn.num_batches_tracked) == 1
    l2 = strat.train_step(b)
    assert math.isfinite(float(l1["total_loss"])) and math.isfinite(float(l2["total_loss"]))


def test_pretrain_point_tower_step_is_bit_reproducible(dev):
    """Stage "pretrain" WITH a point cloud: two identical steps (same batch, same draws, zero learning rate) give bit-equal fp32
    gradient buffers for every unit, the trainable point tower included -- its lga_prep backward is a fixed-order gather since
    round 4 (it was an fp32 atomic scatter; VERDICT r3 weak #1a). Reference: models/mla/pointcloud/backbone/Point_PN.py:115-158."""
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    from mla_amd.strategy import FSDPStrategy
    bb = LLaMa2LLMBackbone(config=LlamaConfig(**recipe.TINY_LLAMA, activation_save_level=1), pad_to_multiple_of=1)
    vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=True, use_contrastive=True, use_generation=False)
    m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=True, use_contrastive=True)
    m.load_state_dict({k: recipe.det_weight(k, v.shape) for k, v in m.state_dict().items()}, strict=True)
    m.freeze_backbones("pretrain")
    strat = FSDPStrategy(vlm=m, device_id=0, stage="pretrain", epochs=1, max_steps=10, global_batch_size=2, per_device_batch_size=2,
                         learning_rate=0.0, weight_decay=0.0, max_grad_norm=1.0, lr_scheduler_type="constant", warmup_ratio=0.0,
                         repeated_diffusion_steps=2)
    strat.run_setup(n_train_examples=20)
    batch, _ = recipe.make_batch(R=2)
    b = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
    b["images"] = {"front_image": batch["images"]["front_image"].to(dev)}
    units = [u for u in strat.sharded.units if u.trainable]
    assert any("vision_tower_3d" in u.name for u in units), [u.name for u in units]
    snaps, norms = [], []
    for _ in range(2):
        torch.manual_seed(7)                     # same noise / timesteps / FPS starts
        strat.train_step(b)
        norms.append(float(strat.sharded._norm))
        snaps.append([u.grad32.clone() for u in units])
    assert norms[0] == norms[1], norms
    for a, c, u in zip(snaps[0], snaps[1], units):
        assert torch.equal(a, c), f"fp32 gradient buffer of unit {u.name} differs between two identical steps"
    tower = [a for a, u in zip(snaps[0], units) if "vision_tower_3d" in u.name][0]
    assert float(tower.abs().max()) > 0
