"""GPU parity of the drop-in modules (HIP path) against the oracle and the reference-captured golden vectors.

Floating-point protocol: the HIP path computes in bf16 with fp32 accumulation, so it is compared with the
fp32 reference (mode A) using the reference's OWN bf16-vs-fp32 spread (mode C vs mode A, both captured from the real
reference) as the yardstick: err(hip, A) <= 2 * err(C, A) + small absolute floor. Integer outputs must be exact."""
import os

import numpy as np
import pytest
import torch

from conftest import fro_rel
from oracle import mla_oracle, recipe
from oracle import torch_oracle as O
from tests_shapes import MLA_TINY_SHAPES

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
BF = torch.bfloat16


@pytest.fixture(scope="module")
def comp():
    return np.load(os.path.join(G, "components.npz"))


@pytest.fixture(scope="module")
def e2e():
    return np.load(os.path.join(G, "mla_tiny_e2e.npz"), allow_pickle=True)


def build_tiny_mla(dev, save_level=2):
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    cfg = LlamaConfig(**recipe.TINY_LLAMA, activation_save_level=save_level)
    bb = LLaMa2LLMBackbone(config=cfg, pad_to_multiple_of=1)
    vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=True, use_contrastive=True,
                       use_generation=False)
    m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=True,
            use_contrastive=True)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == MLA_TINY_SHAPES
    m.load_state_dict(recipe.make_state_dict(MLA_TINY_SHAPES), strict=True)
    m.freeze_backbones("finetune")
    m.train()
    m.to(dev)
    for p in m.parameters():  # bf16 compute weights; BatchNorm running stats stay fp32 (FSDP buffer_dtype)
        p.data = p.data.to(BF)
    return m


@pytest.mark.parametrize("cam", ["rlbench_front", "franka_right", "franka_front"])
def test_projection_exact(dev, comp, cam):
    from mla_amd.fuser import project_points
    idx, valid = project_points(torch.from_numpy(comp["proj_pts"]).to(dev), cam)
    assert np.array_equal(idx.cpu().numpy(), comp[f"proj_idx_{cam}"])
    assert np.array_equal(valid.cpu().numpy(), comp[f"proj_valid_{cam}"])


def test_point_tokenizer_indices_exact_tokens_close(dev, comp):
    from mla_amd.point_tokenizer import PointTokenizer
    pt = PointTokenizer()
    pt.load_state_dict({k: recipe.det_weight("vlm.vision_tower_3d." + k, v.shape) for k, v in pt.state_dict().items()})
    pt.requires_grad_(False).train().to(dev)
    for p in pt.parameters():
        p.data = p.data.to(BF)
    batch, draws = recipe.make_batch(R=1)
    pt.fps_starts_override = [draws["fps_start0"], draws["fps_start1"]]
    tok, ctr = pt(batch["point_cloud"].to(dev))
    (fps0, knn0), (fps1, knn1) = pt.last_indices
    assert np.array_equal(fps0.cpu().numpy(), comp["pt_fps0"]), "stage-1 FPS indices differ from the reference"
    assert np.array_equal(fps1.cpu().numpy(), comp["pt_fps1"]), "stage-2 FPS indices differ from the reference"
    k0 = np.sort(knn0.cpu().numpy(), -1)
    k1 = np.sort(knn1.cpu().numpy(), -1)
    # kNN sets: identical groups expected; a tie at the 81st neighbour may legitimately swap one index
    assert (k0 == comp["pt_knn0_sorted"]).all(-1).mean() > 0.999 and (k1 == comp["pt_knn1_sorted"]).all(-1).mean() > 0.999
    assert np.array_equal(ctr.cpu().numpy(), comp["pt_centers"])
    # tokens: the reference's own bf16-autocast result differs from fp32 by ~1e-1 rel here -> loose bound
    assert fro_rel(tok[:, :, :64], torch.from_numpy(comp["pt_tokens_slice"])) < 0.08
    assert int(pt.patch_embed.EncP.raw_point_embed.net[1].num_batches_tracked) == 1  # BN kept in train mode


def test_vision_tokenizer_tokens(dev, comp):
    from mla_amd.vision_tokenizer import MLP_GELU, VisionTokenizer
    vt = VisionTokenizer(1024)
    vt.load_state_dict({k: recipe.det_weight("vlm.vision_tower_2d." + k, v.shape) for k, v in vt.state_dict().items()})
    proj = MLP_GELU(1024, recipe.TOKEN_SIZE, 2)
    proj.load_state_dict({k: recipe.det_weight("vlm.projector_2d." + k, v.shape) for k, v in proj.state_dict().items()})
    vt.requires_grad_(False).to(dev)
    proj.to(dev)
    for p in list(vt.parameters()) + list(proj.parameters()):
        p.data = p.data.to(BF)
    batch, _ = recipe.make_batch(R=1)
    toks, hw = vt(batch["images"]["front_image"].to(dev), proj)
    got = torch.stack(toks)[:, :, :64]
    assert got.shape == (2, 256, 64) and hw[0].tolist() == [16, 16]
    assert fro_rel(got, torch.from_numpy(comp["vt_tokens_slice"])) < 2e-2
    toks2, _ = vt(batch["images"]["front_image"].repeat(2, 1, 1, 1).to(dev), proj, repeat=2)   # tiled-batch de-dup path
    assert torch.equal(torch.stack(toks2)[:2], torch.stack(toks)) and torch.equal(torch.stack(toks2)[2:], torch.stack(toks))


@pytest.mark.parametrize("case", ["rect_div", "rect_rem", "all_zero", "full"])
def test_vision_tokenizer_cropped_mask_matches_reference(dev, case):
    """a5, the cropped pixel-mask path (models/mla/image/vision_tokenizer.py:124-137) against the reference's own output at B = 1
    (tests/golden/vision_crop.npz from oracle/capture_golden_crop.py): token count [h, w] exact, tokens at the bf16 level of the
    all-ones case above."""
    from mla_amd.vision_tokenizer import MLP_GELU, VisionTokenizer
    from oracle.capture_golden_crop import make_pixels
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vision_crop.npz"))
    vt = VisionTokenizer(1024)
    vt.load_state_dict({k: recipe.det_weight("vlm.vision_tower_2d." + k, v.shape) for k, v in vt.state_dict().items()})
    proj = MLP_GELU(1024, recipe.TOKEN_SIZE, 2)
    proj.load_state_dict({k: recipe.det_weight("vlm.projector_2d." + k, v.shape) for k, v in proj.state_dict().items()})
    vt.requires_grad_(False).to(dev)
    proj.to(dev)
    for p in list(vt.parameters()) + list(proj.parameters()):
        p.data = p.data.to(BF)
    px = make_pixels(case).to(dev)
    toks, hw = vt(px, proj, allow_crop=True)
    assert len(toks) == 1 and hw[0].tolist() == g[f"{case}_hw"].tolist()
    want = torch.from_numpy(g[f"{case}_tokens_slice"])
    assert toks[0].shape[0] == want.shape[0]
    assert fro_rel(toks[0][:, :64], want) < 2e-2
    if case != "full":       # without allow_crop the batched path runs and the owner's deferred check rejects the mask
        vt(px, proj)
        with pytest.raises(NotImplementedError):
            vt.assert_masks_ok()
    # a mixed batch: every sample takes its own rectangle
    both = torch.cat([px, make_pixels("rect_div").to(dev)])
    toks2, hw2 = vt(both, proj, allow_crop=True)
    assert hw2[1].tolist() == [12, 12] and torch.equal(toks2[0], toks[0])


@pytest.mark.parametrize("save_level", [2, 1, 0])
@pytest.mark.parametrize("lens", [None, [100, 37], "odd"])
def test_decoder_layer_fwd_bwd(dev, save_level, lens):
    from mla_amd import ops
    H, I, nh, B, S = 256, 512, 2, 2, 100
    if lens == "odd":            # batch*seq NOT a multiple of 8 (per-device batch 1, odd padded length): zero-row padding path
        B, S, lens = 1, 61, [57]
    names = ["input_layernorm.weight", "self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
             "self_attn.o_proj.weight", "post_attention_layernorm.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
             "mlp.down_proj.weight"]
    shapes = [(H,), (H, H), (H, H), (H, H), (H, H), (H,), (I, H), (I, H), (H, I)]
    p32 = {n: recipe.det_weight("layer." + n, s).to(BF).float() for n, s in zip(names, shapes)}
    x = recipe.det_randn("x", (B, S, H), 1.0).to(BF)
    dy = recipe.det_randn("dy", (B, S, H), 1.0).to(BF)
    seqlens = torch.tensor(lens) if lens else None
    cos, sin = O.rope_tables(S, H // nh)
    xr = x.float().requires_grad_(True)
    pr = {n: v.clone().requires_grad_(True) for n, v in p32.items()}
    ref = O.decoder_layer(xr, pr, cos, sin, nh, 1e-5, seqlens)
    ref.backward(dy.float())
    xd = x.to(dev).requires_grad_(True)
    wd = [p32[n].to(BF).to(dev).requires_grad_(True) for n in names]
    sl = seqlens.to(dev).int() if seqlens is not None else None
    out = ops.decoder_layer(xd, sl, cos.to(dev), sin.to(dev), nh, 1e-5, save_level, wd)
    out.backward(dy.to(dev))
    valid = torch.ones(B, S, dtype=torch.bool) if seqlens is None else torch.arange(S)[None] < seqlens[:, None]
    assert out.shape == (B, S, H) and xd.grad.shape == (B, S, H)
    assert fro_rel(out[valid.to(dev)], ref[valid]) < 1e-2
    assert fro_rel(xd.grad[valid.to(dev)], xr.grad[valid]) < 2e-2
    for n, w in zip(names, wd):
        assert fro_rel(w.grad, pr[n].grad) < 2e-2, n


def test_decoder_layer_at_7b_dimensions(dev):
    """One LlamaDecoderLayer at the benchmark's true dimensions (H 4096, I 11008, 32 heads x 128, S = 548; 2 sequences keep the fp32
    oracle at a few seconds of CPU time): the 256x256 GEMM with its split-K tail, the all-NT backward with its transposes, the
    head-dim-128 attention at the benchmark's sequence length -- output, input gradient and every weight gradient against the oracle."""
    from mla_amd import ops
    H, I, nh, B, S = 4096, 11008, 32, 2, 548
    names = ["input_layernorm.weight", "self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
             "self_attn.o_proj.weight", "post_attention_layernorm.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
             "mlp.down_proj.weight"]
    shapes = [(H,), (H, H), (H, H), (H, H), (H, H), (H,), (I, H), (I, H), (H, I)]
    g = torch.Generator().manual_seed(7)
    p32 = {n: ((torch.ones(s) + 0.1 * torch.randn(s, generator=g)) if len(s) == 1 else 0.02 * torch.randn(s, generator=g)).to(BF).float()
           for n, s in zip(names, shapes)}
    x = torch.randn(B, S, H, generator=g).to(BF)
    dy = torch.randn(B, S, H, generator=g).to(BF)
    seqlens = torch.tensor([S, 500])
    cos, sin = O.rope_tables(S, H // nh)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    xr = x.float().requires_grad_(True)
    pr = {n: v.clone().requires_grad_(True) for n, v in p32.items()}
    ref = O.decoder_layer(xr, pr, cos, sin, nh, 1e-5, seqlens)
    ref.backward(dy.float())
    xd = x.to(dev).requires_grad_(True)
    wd = [p32[n].to(BF).to(dev).requires_grad_(True) for n in names]
    out = ops.decoder_layer(xd, seqlens.to(dev).int(), cos.to(dev), sin.to(dev), nh, 1e-5, 1, wd)
    out.backward(dy.to(dev))
    valid = torch.arange(S)[None] < seqlens[:, None]
    # Yardstick: the SAME oracle run the way the reference runs on the GPU -- bf16 weights and inputs under bf16
    # autocast ("mode C"; every op rounds its result to bf16). The criterion is err(hip, fp32) <= 2 x err(C, fp32) per tensor: the HIP
    # path may not be further from the fp32 truth than twice the reference's own bf16 arithmetic is. north_star's 1e-3 is met per
    # KERNEL on fp32-accumulate outputs (tests/test_kernels_gpu.py: GEMM fp32-out 2e-4); end to end neither side can meet it
    # with bf16 storage (one rounding = 2^-9 per stored tensor, ~6 stored tensors on the longest path).
    xc = x.clone().requires_grad_(True)
    pc = {n: v.to(BF).requires_grad_(True) for n, v in p32.items()}
    with torch.autocast("cpu", dtype=BF):
        refc = O.decoder_layer(xc, pc, cos, sin, nh, 1e-5, seqlens)
    refc.backward(dy)
    errs = {"out": fro_rel(out[valid.to(dev)], ref[valid]), "dx": fro_rel(xd.grad[valid.to(dev)], xr.grad[valid])}
    errc = {"out": fro_rel(refc[valid], ref[valid]), "dx": fro_rel(xc.grad[valid], xr.grad[valid])}
    for n, w in zip(names, wd):
        errs[n] = fro_rel(w.grad, pr[n].grad)
        errc[n] = fro_rel(pc[n].grad, pr[n].grad)
    print("decoder layer @7B dims, Frobenius-relative error vs the fp32 oracle, hip | reference-style bf16 autocast (mode C): " +
          ", ".join(f"{k} {errs[k]:.2e} | {errc[k]:.2e}" for k in errs))
    for k in errs:
        assert errs[k] <= 2.0 * errc[k], (k, errs[k], errc[k])


def test_decoder_layer_config4_recompute_is_bit_identical(dev):
    """BASELINE configs[4] shape on one decoder layer at 7B dimensions (S = 2048, one ragged sequence of two): activation
    checkpointing (save level 0 = recompute the whole forward in the backward, training/strategies/fsdp.py:211-223) must give
    BIT-identical outputs and gradients to keeping every intermediate (level 2) and to the default level 1 -- the recompute runs
    the same kernels on the same inputs, so any difference is a bug in what is saved or re-derived."""
    from mla_amd import ops
    H, I, nh, B, S = 4096, 11008, 32, 2, 2048
    g = torch.Generator().manual_seed(11)
    shapes = [(H,), (H, H), (H, H), (H, H), (H, H), (H,), (I, H), (I, H), (H, I)]
    w32 = [((torch.ones(s) + 0.1 * torch.randn(s, generator=g)) if len(s) == 1 else 0.02 * torch.randn(s, generator=g)) for s in shapes]
    # q|k|v and gate|up contiguous, like the flat unit buffers, so the fused GEMM paths are the ones exercised
    flat = torch.cat([t.reshape(-1) for t in w32]).to(BF).to(dev)
    x = torch.randn(B, S, H, generator=g).to(BF).to(dev)
    dy = torch.randn(B, S, H, generator=g).to(BF).to(dev)
    seqlens = torch.tensor([S, 1777], dtype=torch.int32, device=dev)
    cos, sin = O.rope_tables(S, H // nh)
    cos, sin = cos.to(dev).contiguous(), sin.to(dev).contiguous()
    res = {}
    for lvl in (2, 1, 0):
        ws, off = [], 0
        for sshape in shapes:
            n = int(np.prod(sshape))
            ws.append(flat[off:off + n].view(sshape).detach().requires_grad_(True))
            off += n
        xd = x.clone().requires_grad_(True)
        out = ops.decoder_layer(xd, seqlens, cos, sin, nh, 1e-5, lvl, ws)
        out.backward(dy)
        res[lvl] = (out.detach(), xd.grad, [w.grad for w in ws])
        assert torch.isfinite(out.float()).all() and torch.isfinite(xd.grad.float()).all()
    for lvl in (1, 0):
        assert torch.equal(res[lvl][0], res[2][0]) and torch.equal(res[lvl][1], res[2][1]), lvl
        for a, b in zip(res[lvl][2], res[2][2]):
            assert torch.equal(a, b), lvl
    # ragged sequence: padded rows produce no gradient for their inputs through attention, and the output there is finite
    assert float(res[2][1][1, 1777:].float().abs().max()) < 1e4


@pytest.mark.parametrize("policy", [(0, 1, 3, 2), (3, 3, 0, 0), (1, 0, 1, 0), (0, 0, 0, 0), (3, 3, 3, 3)])
def test_decoder_stack_mixed_activation_policy_is_bit_identical(dev, policy):
    """Mixed activation policy (bench.py --config 4: as many layers as fit in 288 GB keep their activations, the rest are
    checkpointed; training/strategies/fsdp.py:211-223): a 4-layer stack at 7B dimensions, S = 2048, ragged -- every mix of the save
    levels {0 = full recompute, 3 = level 1 without the kept SwiGLU product, 1, 2 = keep all} must give outputs, input gradients and
    all 36 weight gradients BIT-identical to keeping everything."""
    from mla_amd import ops
    H, I, nh, B, S, L = 4096, 11008, 32, 2, 2048, 4
    g = torch.Generator().manual_seed(13)
    shapes = [(H,), (H, H), (H, H), (H, H), (H, H), (H,), (I, H), (I, H), (H, I)]
    flats = []
    for _ in range(L):
        w32 = [((torch.ones(s) + 0.1 * torch.randn(s, generator=g)) if len(s) == 1 else 0.02 * torch.randn(s, generator=g)) for s in shapes]
        flats.append(torch.cat([t.reshape(-1) for t in w32]).to(BF).to(dev))
    x = torch.randn(B, S, H, generator=g).to(BF).to(dev)
    dy = torch.randn(B, S, H, generator=g).to(BF).to(dev)
    seqlens = torch.tensor([S, 1530], dtype=torch.int32, device=dev)
    cos, sin = O.rope_tables(S, H // nh)
    cos, sin = cos.to(dev).contiguous(), sin.to(dev).contiguous()

    def run(levels):
        wall = []
        for flat in flats:
            ws, off = [], 0
            for sshape in shapes:
                n = int(np.prod(sshape))
                ws.append(flat[off:off + n].view(sshape).detach().requires_grad_(True))
                off += n
            wall.append(ws)
        xd = x.clone().requires_grad_(True)
        h = xd
        for ws, lvl in zip(wall, levels):
            h = ops.decoder_layer(h, seqlens, cos, sin, nh, 1e-5, lvl, ws)
        h.backward(dy)
        return h.detach(), xd.grad, [w.grad for ws in wall for w in ws]

    ref = run((2,) * L)
    got = run(policy)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), policy
    for i, (a, b) in enumerate(zip(got[2], ref[2])):
        assert torch.equal(a, b), (policy, i)


def _run_hip_e2e(dev, save_level=2, lazy_lm_head=None):
    m = build_tiny_mla(dev, save_level)
    if lazy_lm_head is not None:
        m.vlm.llm_backbone.llm.config.lazy_lm_head = lazy_lm_head
    batch, draws = recipe.make_batch(R=2)
    m.vlm.vision_tower_3d.fps_starts_override = [draws["fps_start0"], draws["fps_start1"]]
    to = lambda v: v.to(dev)  # noqa: E731
    loss_dict, out = m(input_ids=to(batch["input_ids"]), attention_mask=to(batch["attention_mask"]), labels=to(batch["labels"]),
                       images={"front_image": to(batch["images"]["front_image"])}, point_cloud=to(batch["point_cloud"]),
                       actions=to(batch["actions"]), proprio=to(batch["proprio"]), action_masks=to(batch["action_masks"]),
                       camera_name=batch["camera_name"], repeated_diffusion_steps=2, use_diff=True, noise=to(draws["noise"]),
                       timestep=to(draws["timestep"]))
    loss_dict["total_loss"].backward()
    return m, loss_dict, out


def gradnorm_yardstick(gn, A, C, names):
    """The SURVEY 8c(ii) yardstick for SCALAR gradient norms, with no absolute constant: rel(x) = |x - A| / A per parameter.
    A single parameter's rel(C) can vanish by cancellation (a norm is one number: the reference's own bf16 run hits |C - A| / A = 8e-5
    on final_layer.mlp.fc1.weight while its neighbours sit at 1e-3 .. 4e-2), so the per-parameter bound is twice the LARGER of that
    parameter's own rel(C) and the 90th percentile of rel(C) over all parameters -- the reference's own bf16 spread -- and the medians
    are compared directly. Returns (median rel hip, median rel C, q90 rel C, list of violations)."""
    relA, relC = np.abs(gn - A) / A, np.abs(C - A) / A
    q90 = float(np.quantile(relC, 0.9))
    bad = [(n, float(a), float(c)) for n, a, c in zip(names, relA, relC) if a > 2 * max(c, q90)]
    return float(np.median(relA)), float(np.median(relC)), q90, bad


def test_mla_e2e_against_reference_golden(dev, e2e):
    """Whole tiny-MLA step against the reference golden. Every bound is the yardstick alone -- err(hip, A) <= 2 x err(C, A) with A the
    reference in fp32 and C the reference in its own GPU arithmetic (bf16 autocast) -- no absolute floors (measured
    values per tensor: profiles/r5_parity_table.txt, ratios 0.05 .. 0.76)."""
    m, ld, out = _run_hip_e2e(dev)
    for got, a, c in ((ld["total_loss"], "A_total_loss", "C_total_loss"), (ld["img_pc_contrastive_loss"], "A_contrastive", "C_contrastive"),
                      (out.loss, "A_llm_loss", "C_llm_loss")):
        A, C = float(e2e[a]), float(e2e[c])
        assert abs(float(got) - A) <= 2 * abs(C - A), (a, float(got), A, C)        # measured 1.1e-2 / 4e-5 / 2.2e-3 vs |C - A| 1.4e-2 / 6.9e-4 / 2.8e-2
    assert ld["diff_loss"] is ld["total_loss"]                         # the reference's aliasing quirk (Appendix A #1)
    def err(a, ref):
        return float(np.linalg.norm(a - ref) / np.linalg.norm(ref))
    for name, got in (("hidden8_slice", out.hidden_states[8][:, 250:270, :32]), ("last_hidden_slice", out.hidden_states[-1][:, -8:, :32]),
                      ("logits_slice", out.logits[:, -8:, :64])):
        A, C = e2e["A_" + name], e2e["C_" + name]
        g = got.detach().float().cpu().numpy()
        if name != "hidden8_slice":      # rows -3.. of the ragged samples are padding: flash semantics differ from eager there
            keep = np.ones(A.shape[:2], dtype=bool)
            keep[1, -3:] = keep[3, -3:] = False
            A, C, g = A[keep], C[keep], g[keep]
        assert err(g, A) <= 2 * err(C, A), (name, err(g, A), err(C, A))             # measured ratios 0.05 / 0.43 / 0.45
    # gradients (bf16 .grad tensors here: no main_grad installed)
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    names = [str(n) for n in e2e["grad_names"]]
    assert sorted(grads) == names, "set of parameters receiving gradients differs from the reference"
    gn = np.array([float(grads[k].float().norm()) for k in names])
    medA, medC, q90, bad = gradnorm_yardstick(gn, e2e["A_gradnorms"], e2e["C_gradnorms"], names)
    assert medA <= 2 * medC, (medA, medC)                                            # measured 3.6e-3 vs 3.5e-3
    assert not bad, (q90, bad)                                                       # worst measured 1.8e-2 against 2 x q90 = 4.3e-2
    for key in e2e.files:
        if key.startswith("A_grad::"):
            n = key[len("A_grad::"):]
            A, C = e2e[key], e2e["C_grad::" + n]
            g = grads[n].float().cpu()
            g = (g if tuple(g.shape) == A.shape else g[:16, :64]).numpy()
            assert err(g, A) <= 2 * err(C, A), (n, err(g, A), err(C, A))            # measured ratios 0.15 .. 0.53
    # Round 6: the strict yardstick on EVERY parameter -- err(hip, A) <= 2 x err(C, A) on a 64 x 64 corner (or the
    # whole tensor) of all 116 gradients, not on the 11 captured above; the percentile rule stays for the scalar norms only.
    # Measured: 116 / 116 within 2 x mode C, median ratio 0.44 (profiles/r6_parity_table.txt); no exceptions.
    from parity_util import grad_sample_rows, strict_violations
    rows = grad_sample_rows(grads, e2e)
    assert len(rows) == len(names) == 116
    assert not strict_violations(rows), strict_violations(rows)


def test_lazy_lm_head_is_the_eager_one_on_first_access(dev):
    """Round 6: in training, lm_head(h).float() + the shifted cross entropy (modeling_llama.py:1255-1269) run on
    the first access of output.logits / output.loss instead of inside every forward -- the trainer discards `output`
    (base_strategy_mla.py:307,334). Same kernels on the same hidden states: logits, llm loss (CE + contrastive, added in the reference's
    order), the loss dict and every gradient are BIT-identical to the eager form, and nothing is computed before the first access."""
    res = {}
    for lazy in (True, False):
        m, ld, out = _run_hip_e2e(dev, lazy_lm_head=lazy)
        assert out.lm_head_pending == lazy                                   # after forward + backward nobody has asked yet
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        logits, loss = out.logits, out.loss
        assert not out.lm_head_pending and logits.dtype == torch.float32
        if lazy:
            assert torch.cuda.memory_allocated() - before >= logits.numel() * 4   # the fp32 logits did not exist until now
        assert out.logits is logits and out.loss is loss                     # computed once
        res[lazy] = (logits.detach().cpu(), loss.detach().cpu(), {k: float(v) for k, v in ld.items() if torch.is_tensor(v)},
                     {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None})
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert res[True][2] == res[False][2]
    assert res[True][3].keys() == res[False][3].keys()
    assert all(torch.equal(res[True][3][k], res[False][3][k]) for k in res[True][3])


def test_mla_e2e_against_oracle_flash_semantics(dev):
    """Same step vs the oracle run with flash/varlen pad-row semantics (what the kernels implement)."""
    m, ld, out = _run_hip_e2e(dev, save_level=1)
    sd = recipe.make_state_dict(MLA_TINY_SHAPES)
    batch, draws = recipe.make_batch(R=2)
    with torch.no_grad():
        ref = mla_oracle.mla_forward(sd, batch, draws, 9, 2, 1e-5, 2, zero_pad_rows=True)
    assert abs(float(ld["total_loss"]) - float(ref["total_loss"])) < 5e-2
    assert abs(float(m.last_diff_mse) - float(ref["diff_mse"])) < 2e-2
    last = out.hidden_states[-1].float().cpu()
    pad = ~ref["mask"]
    assert fro_rel(last[~pad], ref["hidden_states"][-1][~pad]) < 3e-2
    assert float(out.hidden_states[5].float().cpu()[pad].abs().max()) < 1e3  # pad rows stay finite


def test_full_size_step_is_deterministic_and_consistent(dev):
    """BASELINE configs[1] at full size (7B, 8 samples x 4 repeats x 548 tokens; what bench.py times): properties that need no CPU
    reference -- the WHOLE step is bit-reproducible (same batch, zero learning rate -> the same loss, the same gradient norm and
    bit-equal fp32 gradient buffers: no floating-point atomics anywhere in the library (round 4: the point tower's lga_prep backward is a
    fixed-order gather too, tests/test_pretrain_gpu.py::test_pretrain_point_tower_step_is_bit_reproducible); the contrastive row gather with repeated targets sums its
    duplicates in a fixed order, ops.GatherRowsSumFn), the global gradient norm equals the norm over the per-unit fp32 gradient buffers, the loss dict carries the reference's
    seven keys with `diff_loss` aliasing `total_loss`, and clipping scales the update (coefficient = 1 / norm for norm > 1)."""
    import math
    import bench
    from mla_amd.strategy import FSDPStrategy
    from mla_amd.synthetic import make_batch
    torch.manual_seed(42)
    mla = bench.build(dev, 1)
    strat = FSDPStrategy(mla, 0, stage="finetune", global_batch_size=8, per_device_batch_size=8, learning_rate=0.0, weight_decay=0.0,
                         max_grad_norm=1.0, lr_scheduler_type="constant", enable_gradient_checkpointing=False, repeated_diffusion_steps=4)
    strat.run_setup(n_train_examples=100)
    batch = make_batch(B=8, L_text=32, seed=42, device=dev, use_pointcloud=True)
    out, snaps = [], []
    probe = [u for u in strat.sharded.units if u.trainable]
    probe = [probe[0], probe[len(probe) // 2], probe[-1]]      # first (projectors / root side), a middle decoder layer, the last unit
    for _ in range(2):
        torch.manual_seed(123)                    # same noise / timesteps / FPS starts in both steps
        ld = strat.train_step(batch)
        out.append((float(ld["total_loss"]), float(strat.sharded._norm), float(strat.sharded._coef)))
        snaps.append([u.grad32.clone() for u in probe])
    assert set(ld) == {"total_loss", "img_pc_contrastive_loss", "tactile_contrastive_loss", "diff_loss", "image_gen_loss",
                       "point_cloud_gen_loss", "tactile_gen_loss"}
    assert float(ld["diff_loss"]) == float(ld["total_loss"])                 # the reference's aliasing (model_mla.py:215-229)
    assert out[0][0] == out[1][0], out                                       # forward: bit-reproducible
    assert out[0][1] == out[1][1], out                                       # backward: bit-equal gradient norm ...
    for a, b, u in zip(snaps[0], snaps[1], probe):
        assert torch.equal(a, b), f"fp32 gradient buffer of unit {u.name} differs between two identical steps"
    del snaps
    loss, norm, coef = out[1]
    assert math.isfinite(loss) and 0.5 < loss < 50.0
    units = [u for u in strat.sharded.units if u.trainable]
    total = math.sqrt(sum(float((u.grad32.double() ** 2).sum()) for u in units))
    assert abs(total - norm) < 1e-3 * norm
    assert abs(coef - min(1.0, 1.0 / (norm + 1e-6))) < 1e-5
    del strat, mla
    torch.cuda.empty_cache()


def test_contrastive_loss_no_valid_correspondence_branch(dev):
    """models/mla/fuser/contrastive.py:206-207: when no point projects into the image (M = 0) the loss is the constant 0.0 with
    requires_grad=True and NO gradient reaches the projection heads or the tapped hidden states; with >= 1 valid pair the same module
    gives a positive loss and gradients. (The reference's train loop reaches this branch whenever valid_mask is all False, e.g. a
    sample whose point cloud lies outside the camera frustum.)"""
    from mla_amd.fuser import CoordinateAwareContrastiveLoss
    torch.manual_seed(3)
    mod = CoordinateAwareContrastiveLoss(feature_dim=128, projection_dim=64).to(dev).to(BF)
    img = torch.randn(2, 16, 128, device=dev).to(BF).requires_grad_(True)
    pc = torch.randn(2, 16, 128, device=dev).to(BF).requires_grad_(True)
    idx = torch.randint(0, 4, (2, 16, 2), device=dev)
    loss0 = mod(img, pc, idx, torch.zeros(2, 16, dtype=torch.bool, device=dev))
    assert float(loss0) == 0.0 and loss0.requires_grad and loss0.grad_fn is None
    loss0.backward()                                              # legal, and touches nothing
    assert img.grad is None and pc.grad is None and all(p.grad is None for p in mod.parameters())
    valid = torch.zeros(2, 16, dtype=torch.bool, device=dev)
    valid[0, 3] = valid[1, 5] = valid[1, 9] = True
    loss = mod(img, pc, idx, valid)
    loss.backward()
    assert float(loss) > 0 and img.grad is not None and float(img.grad.float().abs().max()) > 0
    # oracle on the same weights / inputs (fp32)
    sd = {k: v.detach().float().cpu() for k, v in mod.state_dict().items()}
    heads = {f"{a}_{n}_{wb[0]}": sd[f"{m}_projection_head.{n}.{wb}"] for a, m in (("img", "image"), ("pc", "pointcloud"))
             for n in (0, 2) for wb in ("weight", "bias")}
    ref = O.coordinate_contrastive_loss(img.detach().float().cpu(), pc.detach().float().cpu(), idx.cpu(), valid.cpu(), heads, 0.07)
    assert abs(float(loss) - float(ref)) < 3e-2 * max(1.0, abs(float(ref))), (float(loss), float(ref))
    assert float(O.coordinate_contrastive_loss(img.detach().float().cpu(), pc.detach().float().cpu(), idx.cpu(),
                                               torch.zeros(2, 16, dtype=torch.bool), heads, 0.07)) == 0.0


def test_cropped_pixel_mask_is_rejected(dev):
    """The vision tokenizer only supports the all-ones pixel mask (N_img = 256 layout); the check is deferred to the owner's next
    synchronisation point and must still fire."""
    from mla_amd.vision_tokenizer import MLP_GELU, VisionTokenizer
    vt, proj = VisionTokenizer(1024), MLP_GELU(1024, recipe.TOKEN_SIZE, 2)
    vt.requires_grad_(False).to(dev).to(BF); proj.to(dev).to(BF)
    img = torch.randn(1, 4, 672, 672)
    img[:, 3] = 1.0
    vt(img.to(dev), proj)
    vt.assert_masks_ok()
    img[0, 3, :14] = 0.0
    vt(img.to(dev), proj)
    with pytest.raises(NotImplementedError):
        vt.assert_masks_ok()
