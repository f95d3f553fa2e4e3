"""GPU parity of the tactile path: tactile tokens, TactileContrastiveLoss (batched product + cross entropy),
TactileGenerationModule (single-query decoder) -- whole tiny-MLA step vs the reference golden (tests/golden/mla_tiny_e2e_tactile.npz)
with the C-vs-A yardstick, plus the batched-product op vs autograd."""
import os

import numpy as np
import pytest
import torch

from conftest import fro_rel
from oracle import recipe

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
BF = torch.bfloat16
GEN = dict(use_generation=True, gen_image=False, use_roi=False, gen_pointcloud=False, gen_tactile=True)


def test_bmm_nt_fwd_bwd(dev):
    from mla_amd import ops
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(4, 1, 256, generator=g).to(BF), torch.randn(4, 256, 256, generator=g).to(BF)
    dc = torch.randn(4, 1, 256, generator=g)
    ar, br = a.float().requires_grad_(), b.float().requires_grad_()
    cr = (ar @ br.transpose(1, 2)) / 0.07
    cr.backward(dc)
    ad, bd = a.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    c = ops.BmmNTFn.apply(ad, bd, 1 / 0.07)
    c.backward(dc.to(dev))
    assert fro_rel(c, cr.detach()) < 1e-5
    assert fro_rel(ad.grad, ar.grad) < 1e-2 and fro_rel(bd.grad, br.grad) < 1e-2


def run_tactile_e2e(dev):
    """Builds the tiny MLA with tactile + generation heads, runs forward + backward on the recipe batch; returns (model, loss_dict, golden)."""
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    from test_generation_gpu import _zero_dropout
    gold = np.load(os.path.join(G, "mla_tiny_e2e_tactile.npz"), allow_pickle=True)
    bb = LLaMa2LLMBackbone(config=LlamaConfig(**recipe.TINY_LLAMA, activation_save_level=2), pad_to_multiple_of=1)
    vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=True, use_contrastive=True, use_tactile=True, **GEN)
    m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=True, use_contrastive=True,
            use_tactile=True, **GEN)
    mine = {k: str(tuple(v.shape)) for k, v in m.state_dict().items()}
    assert mine == {str(n): str(s) for n, s in zip(gold["param_names"], gold["param_shapes"])}
    m.load_state_dict({k: recipe.det_weight(k, v.shape) for k, v in m.state_dict().items()}, strict=True)
    m.freeze_backbones("post-training")
    assert "vlm.tactile_embedder" in m.trainable_module_keys and "vlm.tactile_embedder" in m.all_module_keys
    _zero_dropout(m.vlm.generation_manager)
    m.train().to(dev)
    for p in m.parameters():
        p.data = p.data.to(BF)
    batch, draws = recipe.make_batch(R=2, with_tactile=True)
    m.vlm.vision_tower_3d.fps_starts_override = [draws["fps_start0"], draws["fps_start1"]]
    to = lambda v: v.to(dev)  # noqa: E731
    ld, out = m(input_ids=to(batch["input_ids"]), attention_mask=to(batch["attention_mask"]), labels=to(batch["labels"]),
                images={"front_image": to(batch["images"]["front_image"])}, point_cloud=to(batch["point_cloud"]),
                tactile=to(batch["tactile"]), next_tactile=to(batch["next_tactile"]), gripper_xyz=to(batch["gripper_xyz"]),
                actions=to(batch["actions"]), proprio=to(batch["proprio"]), action_masks=to(batch["action_masks"]),
                camera_name=batch["camera_name"], repeated_diffusion_steps=2, use_diff=True, noise=to(draws["noise"]),
                timestep=to(draws["timestep"]))
    ld["total_loss"].backward()
    run_tactile_e2e.last_inputs = (batch, draws)
    return m, ld, gold


def test_mla_e2e_tactile(dev):
    m, ld, gold = run_tactile_e2e(dev)
    batch, draws = run_tactile_e2e.last_inputs
    for key in ("total_loss", "tactile_contrastive_loss", "tactile_gen_loss", "img_pc_contrastive_loss"):
        A, C = float(gold["A_" + key]), float(gold["C_" + key])
        assert abs(float(ld[key]) - A) < 2 * abs(C - A) + 3e-2, (key, float(ld[key]), A, C)
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    names = [str(n) for n in gold["grad_names"]]
    A, C = gold["A_gradnorms"], gold["C_gradnorms"]
    gn = np.array([float(grads[k].float().norm()) if k in grads else 0.0 for k in names])
    live = A > 0
    assert all((k in grads) or not l for k, l in zip(names, live)), [k for k, l in zip(names, live) if l and k not in grads][:5]
    relA, relC = np.abs(gn - A)[live] / A[live], np.abs(C - A)[live] / A[live]
    assert np.median(relA) < 2 * np.median(relC) + 5e-3, (np.median(relA), np.median(relC))
    assert (relA < 2 * relC + 5e-2).mean() > 0.97, [(n, a, c) for n, a, c in zip(np.array(names)[live], relA, relC) if a >= 2 * c + 5e-2][:6]

    def err(a, ref):
        return float(np.linalg.norm(a - ref) / np.linalg.norm(ref))
    for key in gold.files:
        if key.startswith("A_grad::"):
            n = key[len("A_grad::"):]
            Ag, Cg = gold[key], gold["C_grad::" + n]
            g = grads[n].float().cpu()
            got = g.reshape(g.shape[0], -1)[:16, :64].numpy()
            assert err(got, Ag) < 2 * err(Cg, Ag) + 3e-2, (n, err(got, Ag), err(Cg, Ag))
    # Round 6: the strict per-tensor yardstick, no floor, on a gradient sample of EVERY parameter (tests/parity_util.py). Its first run
    # found one tensor at 5.3 x mode C -- the query rows of the tactile decoder's first cross-attention: bf16 probabilities of a
    # near-uniform row do not sum to one, the softmax backward now renormalises them (csrc/gen.hip softmax_rows_bwd_kernel).
    from parity_util import grad_sample_rows, strict_violations
    rows = grad_sample_rows(grads, gold)
    assert len(rows) == len(names)
    # ONE named exception, measured and explained (profiles/r6_parity_table.txt): the query rows of the tactile decoder's first
    # cross-attention. That layer's single learned query attends to ALL LLM states with no key-padding mask, and
    # its query gradient is a covariance over the ~550 keys in which the 3 pad rows of the ragged samples weigh heavily. The golden was
    # captured with the reference's eager attention (pad queries attend to earlier keys); the kernels have the flash / varlen semantics of
    # the reference's GPU path (pad rows -> zero attention output). The fp32 oracle reproduces the golden to 1.8e-6 with eager semantics
    # and sits at 1.25e-1 with flash semantics -- exactly where the HIP path sits (1.21e-1; tools/experiments/dbg_tactile_qgrad.py: an
    # fp32 torch attention core in place of the kernels changes nothing). So this tensor is compared, under the same yardstick, with the
    # oracle run with the kernels' own pad-row semantics.
    KEY = "vlm.generation_manager.tactile_gen_module.decoder.layers.0.multihead_attn.in_proj_weight"
    assert not strict_violations(rows, {KEY}), strict_violations(rows, {KEY})
    from oracle import mla_oracle
    sd = {k: recipe.det_weight(k, v.shape) for k, v in m.state_dict().items()}
    sd[KEY].requires_grad_(True)
    ref = mla_oracle.mla_forward(sd, batch, draws, 9, 2, 1e-5, 2, zero_pad_rows=True, use_tactile=True, gen_tactile=True)
    (gflash,) = torch.autograd.grad(ref["total_loss"], sd[KEY])
    gflash = recipe.grad_slice(gflash).numpy()
    mine = recipe.grad_slice(grads[KEY].float().cpu()).numpy()
    yard = err(gold["C_gs::" + KEY], gold["A_gs::" + KEY])                 # the reference's own bf16 spread on this tensor: 2.3e-2
    assert err(gflash, gold["A_gs::" + KEY]) > 5 * yard                     # the semantics really differ here (else drop the exception)
    assert err(mine, gflash) <= 2 * yard, (err(mine, gflash), yard)
