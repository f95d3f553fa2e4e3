"""Golden vectors for the inference sampler from the REAL reference -- build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/capture_golden_infer.py

1. DDIM-8 / DDPM tables and sampling loops of models/diffusion (create_diffusion, SpacedDiffusion) driven by a closed-form toy
   epsilon model, so the sampler arithmetic is pinned independently of any network.
2. The tiny MLA (vocab 32064 so the ids 29871 / 32001 / 32002 exist) in eval mode: epsilon prediction of one
   PrismaticVLM.forward call and the full 8-step DDIM action chunk, through the same calls predict_action_diff makes
   (model_mla.py:742-752); FPS start indices fixed by feeding torch.randint. NB the shipped predict_action_diff does not
   forward `camera_name` (model_mla.py:726-733), so get_camera_params(None) raises (camera.py:54-56); the capture passes
   camera_name="rlbench_front" through model_kwargs, which is the one-line fix a user of the reference has to make.
Writes tests/golden/inference.npz.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recipe, ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def toy_eps(x, t, scale=0.3, **kw):
    return scale * torch.sin(x * 1.7 + t.float().view(-1, 1, 1) * 0.05) + 0.1 * x


def infer_inputs():
    g = recipe._gen("infer")
    ids = torch.randint(3, 29000, (1, 20), generator=g)
    ids[0, 0] = 1
    ids = torch.cat([ids, torch.tensor([[29871]])], dim=1)         # what predict_action_diff feeds after `[:, :-3]`
    image = torch.cat([torch.randn(1, 3, 672, 672, generator=g), torch.ones(1, 1, 672, 672)], dim=1)
    lo, hi = torch.tensor([0.0, -0.4, 0.75]), torch.tensor([0.6, 0.4, 1.25])
    pc = lo + (hi - lo) * torch.rand(1, 1024, 3, generator=g)
    proprio = torch.rand(1, 1, 7, generator=g) * 2 - 1
    noise = torch.randn(1, 4, 7, generator=g)
    starts = [torch.randint(0, 1024, (1,), generator=g), torch.randint(0, 512, (1,), generator=g)]
    return ids, image, pc, proprio, noise, starts


def main():
    ref_import.setup()
    from models.diffusion import create_diffusion
    res = {}
    d8 = create_diffusion(timestep_respacing="ddim8", noise_schedule="squaredcos_cap_v2", diffusion_steps=100, sigma_small=True, learn_sigma=False)
    res["ddim8_timestep_map"] = np.array(d8.timestep_map)
    res["ddim8_betas"], res["ddim8_acp"], res["ddim8_acp_prev"] = d8.betas, d8.alphas_cumprod, d8.alphas_cumprod_prev
    d10 = create_diffusion(timestep_respacing="ddim10", noise_schedule="squaredcos_cap_v2", diffusion_steps=100, sigma_small=True, learn_sigma=False)
    res["ddim10_timestep_map"] = np.array(d10.timestep_map)
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(2, 4, 7, generator=g)
    res["toy_noise"] = x0.numpy()
    torch.manual_seed(5)
    res["toy_ddim8"] = d8.ddim_sample_loop(toy_eps, x0.shape, x0, clip_denoised=False, model_kwargs={}, progress=False, device="cpu", eta=0.0).numpy()
    torch.manual_seed(5)
    res["toy_ddim8_clip"] = d8.ddim_sample_loop(toy_eps, x0.shape, x0, clip_denoised=True, model_kwargs={}, progress=False, device="cpu", eta=0.0).numpy()
    full = create_diffusion(timestep_respacing="", noise_schedule="squaredcos_cap_v2", diffusion_steps=100, sigma_small=True, learn_sigma=False)
    torch.manual_seed(5)
    res["toy_ddpm100"] = full.p_sample_loop(toy_eps, x0.shape, x0, clip_denoised=False, model_kwargs={}, progress=False, device="cpu").numpy()
    res["post_var"], res["post_logvar"] = full.posterior_variance, full.posterior_log_variance_clipped

    # ---- tiny MLA, eval mode
    cfg = recipe.TINY_LLAMA | {"vocab_size": 32064}
    mla = ref_import.build_reference_mla(cfg, recipe.TOKEN_SIZE, future_action_window_size=3)
    shapes = {k: tuple(v.shape) for k, v in mla.state_dict().items()}
    mla.load_state_dict(recipe.make_state_dict(shapes), strict=True)
    torch.nn.init.normal_(mla.vlm.final_layer.mlp.fc2.weight, std=0.0)          # keep recipe weights (non-zero read-out)
    mla.load_state_dict(recipe.make_state_dict(shapes), strict=True)
    mla.eval()
    ids, image, pc, proprio, noise, starts = infer_inputs()
    up = lambda m, a: tuple(x.float() if torch.is_tensor(x) and x.is_floating_point() else x for x in a)  # noqa: E731
    mla.vlm.proprio_embedder.register_forward_pre_hook(up)
    mla.vlm.x_embedder.register_forward_pre_hook(up)
    o_ri = torch.randint
    seq = []
    torch.randint = lambda *a, **k: seq.pop(0)
    try:
        with torch.no_grad():
            seq[:] = [starts[0], starts[1]]
            out, eps = mla.vlm.forward(noise, torch.tensor([91]), input_ids=ids, images=image, point_cloud=pc, proprio=proprio,
                                       camera_name="rlbench_front")
            res["mla_eps_t91"] = eps.float().numpy()
            res["mla_last_hidden_slice"] = out.hidden_states[-1][:, -8:, :32].float().numpy()
            dd = mla.create_ddim(ddim_step=8)
            seq[:] = [starts[0], starts[1]] * 8
            samples = dd.ddim_sample_loop(mla.vlm.forward, noise.shape, noise, clip_denoised=False,
                                          model_kwargs={"input_ids": ids, "images": image, "point_cloud": pc, "proprio": proprio,
                                                        "camera_name": "rlbench_front"},
                                          progress=False, device="cpu", eta=0.0)
            res["mla_ddim8_actions"] = samples.float().numpy()
    finally:
        torch.randint = o_ri
    res["param_names"] = np.array(sorted(shapes))
    res["param_shapes"] = np.array([str(shapes[k]) for k in sorted(shapes)])
    np.savez_compressed(os.path.join(OUT, "inference.npz"), **res)
    print("inference.npz: ddim8 map", res["ddim8_timestep_map"], "\n eps", res["mla_eps_t91"][0, 0], "\n actions", res["mla_ddim8_actions"][0, 0])


if __name__ == "__main__":
    main()
