"""GPU: the last decoder layer computed on its read-out rows only (round 6, opt-in `MLA.readout_rows_only`; ops.ReadoutLayerFn).

In the diffusion branch the only consumer of the final hidden state is the action read-out (models/vlm/prismatic.py:1115-1126: rows
k + 2 .. k + 2 + T of every sequence -> FinalLayer); lm_head + CE are computed-but-unused and the trainer drops
`output` (training/strategies/base_strategy_mla.py:307,334). A decoder layer is row-wise except for the attention core, so the last layer's
o_proj and MLP -- forward and backward -- run on those rows alone. Checked here: the layer against the dense layer + row gather and
against the fp32 oracle (also at the benchmark's true dimensions under the SURVEY 8c(ii) yardstick), and the whole tiny-MLA step against
the reference golden with the strict per-parameter yardstick; the dense final hidden state / logits are produced on first access."""
import os

import numpy as np
import pytest
import torch

from conftest import fro_rel
from oracle import recipe
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
NAMES = ["input_layernorm.weight", "self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
         "self_attn.o_proj.weight", "post_attention_layernorm.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
         "mlp.down_proj.weight"]


def _params(H, I, seed=None):
    shapes = [(H,), (H, H), (H, H), (H, H), (H, H), (H,), (I, H), (I, H), (H, I)]
    if seed is None:
        return {n: recipe.det_weight("layer." + n, s).to(BF).float() for n, s in zip(NAMES, shapes)}
    g = torch.Generator().manual_seed(seed)
    return {n: ((torch.ones(s) + 0.1 * torch.randn(s, generator=g)) if len(s) == 1 else 0.02 * torch.randn(s, generator=g)).to(BF).float()
            for n, s in zip(NAMES, shapes)}


def _packed(p32, dev):
    """The nine weights as views of one flat bf16 leaf (FlatUnit layout: q|k|v and gate|up back to back)."""
    offs, n = {}, 0
    for name in NAMES:
        offs[name] = (n, tuple(p32[name].shape))
        n += p32[name].numel()
    flat = torch.empty(n, dtype=BF, device=dev)
    for name in NAMES:
        o, _ = offs[name]
        flat[o:o + p32[name].numel()] = p32[name].to(BF).to(dev).reshape(-1)
    flat.requires_grad_(True)
    views = [flat[offs[name][0]:offs[name][0] + p32[name].numel()].view(offs[name][1]) for name in NAMES]
    return views, flat, offs


def _g(flat, offs, name):
    o, shp = offs[name]
    return flat.grad[o:o + int(np.prod(shp))].view(shp)


@pytest.mark.parametrize("case", ["one_row", "two_rows_ragged", "pad"])
def test_readout_layer_is_the_dense_layer_on_the_read_rows(dev, case):
    """ReadoutLayerFn == DecoderLayerFn followed by a row gather: output rows, input gradient and every weight gradient agree with the
    dense layer (given d(out) = zero outside the read rows) to rounding, and with the fp32 oracle within the decoder-layer bounds.
    Cases: one read row per sequence; two rows per sequence with right-padded sequences; a token count that gets zero rows appended."""
    from mla_amd import ops
    H, I, nh, B, S = 256, 512, 2, 3, 100
    lens, per = None, 1
    if case == "two_rows_ragged":
        lens, per = [100, 61, 80], 2
    if case == "pad":
        S = 107
    p32 = _params(H, I)
    x = recipe.det_randn("x", (B, S, H), 1.0).to(BF)
    seqlens = torch.tensor(lens) if lens else None
    last = torch.tensor(lens) if lens else torch.full((B,), S)
    rows = torch.cat([torch.arange(per) + b * S + int(last[b]) - 2 - per for b in range(B)])          # near the end of every valid prefix
    n = rows.numel()
    dy = recipe.det_randn("dy", (n, H), 1.0).to(BF)
    cos, sin = O.rope_tables(S, H // nh)
    # fp32 oracle of the dense layer, gradient injected on the read rows only
    xr = x.float().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in p32.items()}
    ref = O.decoder_layer(xr, pr, cos, sin, nh, 1e-5, seqlens)
    ref.reshape(B * S, H)[rows].backward(dy.float())
    sl = seqlens.to(dev).int() if seqlens is not None else None
    # dense HIP layer + gather
    xd = x.to(dev).requires_grad_(True)
    wd, flat_d, offs = _packed(p32, dev)
    dense = ops.decoder_layer(xd, sl, cos.to(dev), sin.to(dev), nh, 1e-5, 1, wd)
    dense.reshape(B * S, H)[rows.to(dev)].backward(dy.to(dev))
    # read-out layer
    xs = x.to(dev).requires_grad_(True)
    ws, flat_s, _ = _packed(p32, dev)
    out = ops.decoder_layer_readout(xs, sl, cos.to(dev), sin.to(dev), nh, 1e-5, rows.to(dev), ws)
    assert out.shape == (n, H)
    out.backward(dy.to(dev))
    want = dense.detach().reshape(B * S, H)[rows.to(dev)]
    assert fro_rel(out, want) < 6e-3 and fro_rel(out, ref.reshape(B * S, H)[rows]) < 1e-2
    valid = torch.ones(B, S, dtype=torch.bool) if seqlens is None else torch.arange(S)[None] < seqlens[:, None]
    assert fro_rel(xs.grad[valid.to(dev)], xd.grad[valid.to(dev)]) < 1.5e-2
    assert fro_rel(xs.grad[valid.to(dev)], xr.grad[valid]) < 2e-2
    for name in NAMES:
        a, b, c = _g(flat_s, offs, name), _g(flat_d, offs, name), pr[name].grad
        assert fro_rel(a, b) < 1.5e-2, (name, fro_rel(a, b))
        assert fro_rel(a, c) < 2e-2, (name, fro_rel(a, c))
    # run to run: bit-identical
    xs2 = x.to(dev).requires_grad_(True)
    ws2, flat_s2, _ = _packed(p32, dev)
    out2 = ops.decoder_layer_readout(xs2, sl, cos.to(dev), sin.to(dev), nh, 1e-5, rows.to(dev), ws2)
    out2.backward(dy.to(dev))
    assert torch.equal(out, out2) and torch.equal(xs.grad, xs2.grad) and torch.equal(flat_s.grad, flat_s2.grad)


def test_readout_layer_at_7b_dimensions(dev):
    """The read-out layer at the benchmark's true dimensions (H 4096, I 11008, 32 heads, S = 548, one read row per sequence) under the
    yardstick alone: err(hip, fp32 oracle) <= 2 x err(reference-style bf16 autocast, fp32 oracle) on the read rows, the input gradient and
    every weight gradient."""
    from mla_amd import ops
    H, I, nh, B, S = 4096, 11008, 32, 2, 548
    p32 = _params(H, I, seed=7)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, S, H, generator=g).to(BF)
    seqlens = torch.tensor([S, 500])
    rows = torch.tensor([S - 3, S + 500 - 3])
    dy = torch.randn(2, H, generator=g).to(BF)
    cos, sin = O.rope_tables(S, H // nh)
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    xr = x.float().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in p32.items()}
    ref = O.decoder_layer(xr, pr, cos, sin, nh, 1e-5, seqlens)
    ref.reshape(B * S, H)[rows].backward(dy.float())
    xc = x.clone().requires_grad_(True)
    pc = {k: v.to(BF).requires_grad_(True) for k, v in p32.items()}
    with torch.autocast("cpu", dtype=BF):
        refc = O.decoder_layer(xc, pc, cos, sin, nh, 1e-5, seqlens)
    refc.reshape(B * S, H)[rows].backward(dy)
    xs = x.to(dev).requires_grad_(True)
    ws, flat, offs = _packed(p32, dev)
    out = ops.decoder_layer_readout(xs, seqlens.to(dev).int(), cos.to(dev), sin.to(dev), nh, 1e-5, rows.to(dev), ws)
    out.backward(dy.to(dev))
    valid = torch.arange(S)[None] < seqlens[:, None]
    errs = {"out rows": fro_rel(out, ref.reshape(B * S, H)[rows]), "dx": fro_rel(xs.grad[valid.to(dev)], xr.grad[valid])}
    errc = {"out rows": fro_rel(refc.reshape(B * S, H)[rows], ref.reshape(B * S, H)[rows]), "dx": fro_rel(xc.grad[valid], xr.grad[valid])}
    for name in NAMES:
        errs[name] = fro_rel(_g(flat, offs, name), pr[name].grad)
        errc[name] = fro_rel(pc[name].grad, pr[name].grad)
    line = ("READ-OUT decoder layer @7B dims (2 read rows), Frobenius-relative error vs the fp32 oracle, hip | reference-style bf16 autocast (mode C): " +
            ", ".join(f"{k} {errs[k]:.2e} | {errc[k]:.2e}" for k in errs))
    print(line)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_readout_layer_7b.txt", "w") as f:
        f.write(line + "\n")
    for k in errs:
        assert errs[k] <= 2.0 * errc[k], (k, errs[k], errc[k])


def test_tiny_mla_step_with_readout_rows_against_the_reference_golden(dev):
    """The whole tiny-MLA training step with MLA.readout_rows_only against the reference golden: the losses and the strict per-parameter
    yardstick err(hip, A) <= 2 x err(C, A) on all 116 gradients (the bounds of test_mla_e2e_against_reference_golden). The dense final
    hidden state and the logits are NOT computed by the step; they appear on first access and agree with the golden / the read rows."""
    import test_model_gpu as tm
    from mla_amd import ops
    from parity_util import grad_sample_rows, strict_violations
    e2e = np.load(os.path.join(tm.G, "mla_tiny_e2e.npz"), allow_pickle=True)
    orig_build = tm.build_tiny_mla
    calls = []
    orig_fn = ops.decoder_layer_readout

    def build(dev_, save_level=2):
        m = orig_build(dev_, save_level)
        m.readout_rows_only = True
        return m

    tm.build_tiny_mla = build
    ops.decoder_layer_readout = lambda *a, **k: (calls.append(1), orig_fn(*a, **k))[1]
    try:
        m, ld, out = tm._run_hip_e2e(dev)
    finally:
        tm.build_tiny_mla = orig_build
        ops.decoder_layer_readout = orig_fn
    assert calls == [1], calls                                   # the last layer, once
    assert out.last_hidden_pending and out.lm_head_pending       # nothing dense was computed by the step
    for got, a, c in ((ld["total_loss"], "A_total_loss", "C_total_loss"), (ld["img_pc_contrastive_loss"], "A_contrastive", "C_contrastive")):
        A, C = float(e2e[a]), float(e2e[c])
        assert abs(float(got) - A) <= 2 * abs(C - A), (a, float(got), A, C)
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    names = [str(n) for n in e2e["grad_names"]]
    assert sorted(grads) == names
    rows = grad_sample_rows(grads, e2e)
    assert len(rows) == 116
    assert not strict_violations(rows), strict_violations(rows)

    def err(a, ref):
        return float(np.linalg.norm(a - ref) / np.linalg.norm(ref))
    # first access: the dense last layer + final norm run now (9 layer outputs + the final state), and the logits / llm loss after them
    hs = out.hidden_states
    assert not out.last_hidden_pending and len(hs) == 10
    A, C = e2e["A_last_hidden_slice"], e2e["C_last_hidden_slice"]
    g = hs[-1][:, -8:, :32].detach().float().cpu().numpy()
    keep = np.ones(A.shape[:2], dtype=bool)
    keep[1, -3:] = keep[3, -3:] = False
    assert err(g[keep], A[keep]) <= 2 * err(C[keep], A[keep])
    A, C = float(e2e["A_llm_loss"]), float(e2e["C_llm_loss"])
    assert abs(float(out.loss) - A) <= 2 * abs(C - A)
    assert not out.lm_head_pending


def test_llama_stack_with_readout_rows_and_folded_norms_together(dev):
    """LlamaModel.forward(readout_rows=...) with MLA_NORM_FOLD on as well (both opt-ins): the layer in front of the last one must NOT
    prepare a folded input norm for the read-out layer (it runs the plain rmsnorm), the read rows equal the dense stack's rows of the
    final hidden state to rounding, and so do the input gradient and every parameter gradient."""
    from mla_amd import ops
    from mla_amd.llama import LlamaConfig, LlamaModel
    cfg = LlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
                      rms_norm_eps=1e-5, activation_save_level=1)
    torch.manual_seed(5)
    m = LlamaModel(cfg).to(dev).to(BF)
    for layer in m.layers:
        ps = list(layer._weights())
        flat = torch.empty(sum(p.numel() for p in ps), dtype=BF, device=dev)
        o = 0
        for p in ps:
            flat[o:o + p.numel()] = p.data.reshape(-1)
            p.data = flat[o:o + p.numel()].view(p.shape)
            o += p.numel()
    B, S, H = 3, 128, 256
    emb = recipe.det_randn("emb", (B, S, H), 1.0).to(BF).to(dev)
    rows = torch.tensor([100, S + 90, 2 * S + 127], device=dev)
    dy = recipe.det_randn("dy", (3, H), 1.0).to(BF).to(dev)

    def run(readout, fold):
        for p in m.parameters():
            p.grad = None
        prev = ops.set_norm_fold(fold)
        try:
            x = emb.clone().requires_grad_(True)
            if readout:
                picked, hidden, dense_last = m(inputs_embeds=x, output_hidden_states=True, readout_rows=rows)
                assert len(hidden) == 3 and callable(dense_last)
            else:
                last, _ = m(inputs_embeds=x)
                picked = last.reshape(B * S, H)[rows]
            picked.backward(dy)
        finally:
            ops.set_norm_fold(prev)
        return picked.detach(), x.grad, {n: p.grad for n, p in m.named_parameters() if p.grad is not None}

    a = run(True, True)
    b = run(False, False)
    assert fro_rel(a[0], b[0]) < 1e-2 and fro_rel(a[1], b[1]) < 2.5e-2
    assert a[2].keys() == b[2].keys()
    for n in a[2]:
        assert fro_rel(a[2][n], b[2][n]) < 2.5e-2, (n, fro_rel(a[2][n], b[2][n]))
    a2 = run(True, True)
    assert torch.equal(a[0], a2[0]) and torch.equal(a[1], a2[1])
