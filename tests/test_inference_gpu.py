"""GPU parity of the inference path: eval-mode PrismaticVLM.forward as the epsilon model and the 8-step DDIM
action sampler (MLA.predict_action_diff) against vectors captured from the real reference (tests/golden/inference.npz, fp32).
The HIP path is bf16: epsilon within 5e-2 (Frobenius-relative), the 8-step chunk within 1e-1 (errors compound over the steps)."""
import os

import numpy as np
import pytest
import torch

from conftest import fro_rel
from oracle import recipe

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
BF = torch.bfloat16


def infer_inputs():
    """Same recipe as oracle/capture_golden_infer.py:infer_inputs."""
    g = recipe._gen("infer")
    ids = torch.randint(3, 29000, (1, 20), generator=g)
    ids[0, 0] = 1
    ids = torch.cat([ids, torch.tensor([[29871]])], dim=1)
    image = torch.cat([torch.randn(1, 3, 672, 672, generator=g), torch.ones(1, 1, 672, 672)], dim=1)
    lo, hi = torch.tensor([0.0, -0.4, 0.75]), torch.tensor([0.6, 0.4, 1.25])
    pc = lo + (hi - lo) * torch.rand(1, 1024, 3, generator=g)
    proprio = torch.rand(1, 1, 7, generator=g) * 2 - 1
    noise = torch.randn(1, 4, 7, generator=g)
    starts = [torch.randint(0, 1024, (1,), generator=g), torch.randint(0, 512, (1,), generator=g)]
    return ids, image, pc, proprio, noise, starts


@pytest.fixture(scope="module")
def model(dev):
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    gold = np.load(os.path.join(G, "inference.npz"), allow_pickle=True)
    bb = LLaMa2LLMBackbone(config=LlamaConfig(**(recipe.TINY_LLAMA | {"vocab_size": 32000})))      # + <PAD>, padded to 32064 rows
    vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=True, use_contrastive=True,
                       use_generation=False, future_action_window_size=3)
    m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=3, use_diff=True, use_pointcloud=True, use_contrastive=True)
    mine = {k: str(tuple(v.shape)) for k, v in m.state_dict().items()}
    assert mine == {str(n): str(s) for n, s in zip(gold["param_names"], gold["param_shapes"])}
    m.load_state_dict({k: recipe.det_weight(k, v.shape) for k, v in m.state_dict().items()}, strict=True)
    m.eval().to(dev)
    for p in m.parameters():
        p.data = p.data.to(BF)
    ids, image, pc, proprio, noise, starts = infer_inputs()
    m.vlm.vision_tower_3d.fps_starts_override = starts
    return m, gold


def test_eval_forward_epsilon(dev, model):
    m, gold = model
    ids, image, pc, proprio, noise, _ = infer_inputs()
    with torch.inference_mode():
        out, eps = m.vlm(noise.to(dev), torch.tensor([91], device=dev), input_ids=ids.to(dev), images=image.to(dev), point_cloud=pc.to(dev),
                         proprio=proprio.to(dev), camera_name="rlbench_front")
    assert fro_rel(out.hidden_states[-1][:, -8:, :32], torch.from_numpy(gold["mla_last_hidden_slice"])) < 3e-2
    assert fro_rel(eps, torch.from_numpy(gold["mla_eps_t91"])) < 5e-2
    assert out.loss is None or not torch.is_tensor(out.loss) or out.loss.numel() <= 1


def test_predict_action_diff_ddim8(dev, model):
    m, gold = model
    ids, image, pc, proprio, noise, _ = infer_inputs()
    act = m.predict_action_diff(image=image[0], pointcloud=pc[0].numpy(), cur_robot_state=proprio[0, 0].numpy(), input_ids=ids,
                                noise=noise, num_ddim_steps=8, use_ddim=True)
    assert act.shape == (4, 7) and m.ddim_diffusion.timestep_map == [0, 13, 26, 39, 52, 65, 78, 91]
    ref = gold["mla_ddim8_actions"][0]
    assert np.linalg.norm(act - ref) / np.linalg.norm(ref) < 1e-1, (act[0], ref[0])
    # the un-tailed prompt form (ids without the trailing 29871) gets the reference's tail appended and gives the same chunk
    act2 = m.predict_action_diff(image=image[0, :3], pointcloud=pc[0], cur_robot_state=proprio[0, 0].numpy(), input_ids=ids[:, :-1],
                                 noise=noise, num_ddim_steps=8)
    assert np.allclose(act, act2, rtol=1e-5, atol=1e-6)


def test_unnormalisation_and_latency(dev, model):
    m, _ = model
    ids, image, pc, proprio, noise, _ = infer_inputs()
    q01, q99 = np.linspace(-0.5, -0.1, 7), np.linspace(0.2, 1.0, 7)
    m.norm_stats = {"rlbench": {"action": {"q01": q01.tolist(), "q99": q99.tolist(), "mask": [True] * 6 + [False]},
                                "proprio": {"q01": (-np.ones(7)).tolist(), "q99": np.ones(7).tolist()}}}
    try:
        raw = m.predict_action_diff(image=image[0], pointcloud=pc[0], cur_robot_state=proprio[0, 0].numpy(), input_ids=ids, noise=noise)
        m2_stats = m.norm_stats
        m.norm_stats = None
        norm = m.predict_action_diff(image=image[0], pointcloud=pc[0], cur_robot_state=proprio[0, 0].numpy(), input_ids=ids, noise=noise)
        m.norm_stats = m2_stats
        a = np.clip(norm, -1, 1)
        a[:, 6] = np.where(a[:, 6] < 0.5, 0, 1)
        want = np.where(np.array([True] * 6 + [False]), 0.5 * (a + 1) * (q99 - q01) + q01, a)
        assert np.allclose(raw, want, rtol=1e-5, atol=1e-6)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(3):
            m.predict_action_diff(image=image[0], pointcloud=pc[0], cur_robot_state=proprio[0, 0].numpy(), input_ids=ids, noise=noise)
        torch.cuda.synchronize()
        print(f"tiny-model 8-step DDIM latency: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms")
    finally:
        m.norm_stats = None


def test_predict_action_diff_from_uint8_frame(dev, model):
    """A raw 224x224 uint8 frame goes through the PIL-exact GPU preprocessing (CLIPImageProcessor step of model_mla.py:656-660)."""
    m, _ = model
    ids, _, pc, proprio, noise, _ = infer_inputs()
    frame = np.random.RandomState(3).randint(0, 256, (224, 224, 3)).astype(np.uint8)
    a1 = m.predict_action_diff(image=frame, pointcloud=pc[0], cur_robot_state=proprio[0, 0].numpy(), input_ids=ids, noise=noise)
    pv = m.vlm.get_vision_tower_2d().image_processor.preprocess(frame)["pixel_values"][0]
    a2 = m.predict_action_diff(image=pv, pointcloud=pc[0], cur_robot_state=proprio[0, 0].numpy(), input_ids=ids, noise=noise)
    assert a1.shape == (4, 7) and np.allclose(a1, a2, rtol=1e-5, atol=1e-6)


def test_predict_action_diff_from_instruction(dev, model):
    """`instruction` path (model_mla.py:626-645): prompt from the backbone's builder, ids from the attached tokenizer, the
    [29871, 32001, 32002, 29871] tail appended and its last three ids dropped == passing those ids directly."""
    m, _ = model
    ids, image, pc, proprio, noise, _ = infer_inputs()
    seen = {}

    class Tok:
        def __call__(self, text, truncation=True, return_tensors="pt"):
            seen["text"] = text
            return type("Enc", (), {"input_ids": ids[:, :-1].clone()})()
    bb = m.vlm.llm_backbone
    old_tok, old_id = getattr(bb, "tokenizer", None), bb.identifier
    bb.tokenizer, bb.identifier = Tok(), "llama2-7b-pure"
    try:
        a1 = m.predict_action_diff(image=image[0], pointcloud=pc[0], instruction="Close The JAR", cur_robot_state=proprio[0, 0].numpy(),
                                   noise=noise)
    finally:
        bb.tokenizer, bb.identifier = old_tok, old_id
    assert seen["text"] == "In: What action should the robot take to close the jar?\nOut:"
    a2 = m.predict_action_diff(image=image[0], pointcloud=pc[0], cur_robot_state=proprio[0, 0].numpy(), input_ids=ids, noise=noise)
    assert np.allclose(a1, a2, rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        m.predict_action_diff(image=image[0], pointcloud=pc[0], cur_robot_state=proprio[0, 0].numpy(), noise=noise)
