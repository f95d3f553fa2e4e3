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


def test_mla_e2e_pretrain_stage(dev):
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    gold = np.load(os.path.join(G, "mla_tiny_e2e_pretrain.npz"), allow_pickle=True)
    bb = LLaMa2LLMBackbone(config=LlamaConfig(**recipe.TINY_LLAMA, activation_save_level=2), pad_to_multiple_of=1)
    vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=False, use_contrastive=False, use_generation=False)
    m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=False, use_contrastive=False)
    mine = {k: str(tuple(v.shape)) for k, v in m.state_dict().items()}
    assert mine == {str(n): str(s) for n, s in zip(gold["param_names"], gold["param_shapes"])}
    m.load_state_dict({k: recipe.det_weight(k, v.shape) for k, v in m.state_dict().items()}, strict=True)
    m.freeze_backbones("pretrain")
    m.train().to(dev)
    for p in m.parameters():
        p.data = p.data.to(BF)
    batch, draws = recipe.make_batch(R=2)
    to = lambda v: v.to(dev)  # noqa: E731
    ld, out = m(input_ids=to(batch["input_ids"]), attention_mask=to(batch["attention_mask"]), labels=to(batch["labels"]),
                images={"front_image": to(batch["images"]["front_image"])}, actions=to(batch["actions"]), proprio=to(batch["proprio"]),
                action_masks=to(batch["action_masks"]), camera_name=batch["camera_name"], repeated_diffusion_steps=2, use_diff=True,
                noise=to(draws["noise"]), timestep=to(draws["timestep"]))
    ld["total_loss"].backward()
    tolL = 2 * abs(float(gold["C_total_loss"]) - float(gold["A_total_loss"])) + 2e-2
    assert abs(float(ld["total_loss"]) - float(gold["A_total_loss"])) < tolL
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    names = [str(n) for n in gold["grad_names"]]
    assert sorted(grads) == names, sorted(set(names) ^ set(grads))
    gn = np.array([float(grads[k].float().norm()) for k in names])
    relA = np.abs(gn - gold["A_gradnorms"]) / (gold["A_gradnorms"] + 1e-12)
    relC = np.abs(gold["C_gradnorms"] - gold["A_gradnorms"]) / (gold["A_gradnorms"] + 1e-12)
    assert np.median(relA) < 2 * np.median(relC) + 5e-3, (np.median(relA), np.median(relC))
    assert (relA < 2 * relC + 5e-2).mean() > 0.97, [(n, a, c) for n, a, c in zip(names, relA, relC) if a >= 2 * c + 5e-2][:6]

    def err(a, ref):
        return float(np.linalg.norm(a - ref) / np.linalg.norm(ref))
    for key in gold.files:
        if key.startswith("A_grad::"):
            n = key[len("A_grad::"):]
            A, C = gold[key], gold["C_grad::" + n]
            g = grads[n].float().cpu()
            got = (g.reshape(g.shape[0], -1)[:16, :64] if A.ndim == 2 else g.reshape(-1)[:256]).numpy()
            assert err(got, A) < 2 * err(C, A) + 3e-2, (n, err(got, A), err(C, A))
