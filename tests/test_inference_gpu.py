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


# ------------------------------------------------------------------------------------------------ prefix reuse (round 6, mla_amd/infer.py)
@pytest.mark.parametrize("M,N,K,res", [(1, 4096, 4096, True), (2, 12288, 4096, False), (5, 1000, 11008, True), (8, 22016, 4096, False), (3, 7, 512, False)])
def test_gemv_matches_fp32_reference(dev, M, N, K, res):
    """mla_gemv_bf16: out[m] = x[m] @ W^T (+ residual) for M <= 8 rows, fp32 accumulation, incl. N not a multiple of the 4 rows a wave
    keeps in flight, K not a multiple of 512, and the (rows per sample, sample stride) output addressing the cache slots use."""
    from mla_amd import hip
    g = torch.Generator().manual_seed(M * 1000 + N)
    x = (torch.randn(M, K, generator=g) * 0.5).to(BF)
    W = (torch.randn(N, K, generator=g) * 0.05).to(BF)
    r = (torch.randn(M, N, generator=g)).to(BF) if res else None
    want = x.float() @ W.float().t() + (r.float() if res else 0)
    out = torch.full((M, N), float("nan"), dtype=BF, device=dev)
    hip.gemv(x.to(dev), W.to(dev), out, N, 0, M, r.to(dev) if res else None)
    assert torch.isfinite(out.float()).all()
    assert fro_rel(out, want) < 4e-3                                           # one bf16 rounding of the fp32 sums
    if M % 2 == 0 and N % 8 == 0:
        # rows of sample b land at out + b * batch_stride + r * ldo (+ column offset): the q|k|v cache addressing
        rpb, S_cap, ld = 2, 5, N + 64
        buf = torch.zeros((M // rpb, S_cap, ld), dtype=BF, device=dev)
        hip.gemv(x.to(dev), W.to(dev), buf[:, 3:], ld, buf.stride(0), rpb, None, out_col=32)
        got = buf[:, 3:5, 32:32 + N].reshape(M, N)
        ref = torch.full((M, N), float("nan"), dtype=BF, device=dev)
        hip.gemv(x.to(dev), W.to(dev), ref, N, 0, M, None)
        assert torch.equal(got, ref)
        assert float(buf[:, :3].float().abs().max()) == 0 and float(buf[:, 3:, :32].float().abs().max()) == 0 and float(buf[:, 3:, 32 + N:].float().abs().max()) == 0


@pytest.mark.parametrize("M,K,N", [(2, 4096, 512), (5, 256, 96), (7, 8192, 64)])
def test_gemv_fused_rmsnorm_and_swiglu_inputs_match_the_separate_kernels(dev, M, K, N):
    """mla_gemv_bf16 with pre = 1 / 2: LlamaRMSNorm resp. SwiGLU applied to the rows inside the kernel's input staging must be the
    stand-alone kernels' arithmetic -- bit-identical outputs to rmsnorm_fwd / swiglu_fwd followed by the plain GEMV."""
    from mla_amd import hip
    g = torch.Generator().manual_seed(K + M)
    x = (torch.randn(M, K, generator=g) * 1.3).to(BF).to(dev)
    w = (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
    gu = torch.randn(M, 2 * K, generator=g).to(BF).to(dev)
    a, b = (torch.full((M, N), float("nan"), dtype=BF, device=dev) for _ in range(2))
    hip.gemv(hip.rmsnorm_fwd(x, w, 1e-5)[0], W, a, N, 0, M)
    hip.gemv(x, W, b, N, 0, M, norm_weight=w, eps=1e-5)
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)
    a.fill_(float("nan")); b.fill_(float("nan"))
    hip.gemv(hip.swiglu_fwd(gu), W, a, N, 0, M)
    hip.gemv(gu, W, b, N, 0, M, swiglu=True)
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)


@pytest.mark.parametrize("B,R,nh", [(1, 2, 32), (2, 4, 2), (1, 8, 3)])
def test_gemv_fused_rmsnorm_rope_qkv_is_the_three_kernels(dev, B, R, nh):
    """north_star's "fused RMSNorm + RoPE + QKV" as ONE kernel on the inference path: mla_gemv_bf16 with the RMSNorm in its input
    staging and the rotary embedding of the q | k columns in its epilogue, writing into per-sample cache slots -- bit-identical to
    rmsnorm_fwd + plain GEMV + rope_inplace on the same rows (positions S_p .. S_p + R - 1)."""
    from mla_amd import hip
    D, K = 128, 512
    H = nh * D
    M, S_p, S_cap = B * R, 11, 11 + R
    g = torch.Generator().manual_seed(nh * 10 + R)
    x = (torch.randn(M, K, generator=g) * 1.1).to(BF).to(dev)
    w = (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(dev)
    W = (torch.randn(3 * H, K, generator=g) * 0.06).to(BF).to(dev)
    pos = torch.arange(S_p, S_cap).float()
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    fr = torch.outer(pos, inv)
    cos, sin = fr.cos().contiguous().to(dev), fr.sin().contiguous().to(dev)
    ref = torch.zeros((B, S_cap, 3 * H), dtype=BF, device=dev)
    hip.gemv(hip.rmsnorm_fwd(x, w, 1e-5)[0], W, ref[:, S_p:], 3 * H, ref.stride(0), R)
    for b in range(B):
        hip.rope_inplace(ref[b, S_p:], cos, sin, R, nh, D, 0, H)
    got = torch.zeros_like(ref)
    hip.gemv(x, W, got[:, S_p:], 3 * H, got.stride(0), R, norm_weight=w, eps=1e-5, rope=(cos, sin, 2 * H))
    assert torch.isfinite(got.float()).all() and float(got[:, :S_p].float().abs().max()) == 0
    assert torch.equal(got, ref)
    assert not torch.equal(got[:, S_p:, :2 * H], hip_plain(x, w, W, B, R, S_p, S_cap, H, dev)[:, S_p:, :2 * H])    # the rotation really happened


def hip_plain(x, w, W, B, R, S_p, S_cap, H, dev):
    from mla_amd import hip
    out = torch.zeros((B, S_cap, 3 * H), dtype=BF, device=dev)
    hip.gemv(x, W, out[:, S_p:], 3 * H, out.stride(0), R, norm_weight=w, eps=1e-5)
    return out


@pytest.mark.parametrize("B,H,S_kv,R", [(1, 32, 550, 2), (2, 4, 77, 5), (1, 2, 8, 8), (3, 3, 1030, 1)])
def test_attn_decode_matches_fp32_reference(dev, B, H, S_kv, R):
    """mla_attn_decode: the last R rows of the packed q|k|v cache are the queries; query r attends to keys [0, S_kv - R + r]."""
    import math
    from mla_amd import hip
    D = 128
    g = torch.Generator().manual_seed(S_kv + R)
    cache = (torch.randn(B, S_kv + 3, 3 * H * D, generator=g) * 0.7).to(BF)
    o = hip.attn_decode(cache.to(dev), B, H, D, S_kv, R, 1 / math.sqrt(D))
    c = cache.float()[:, :S_kv]
    q = c[:, S_kv - R:, :H * D].view(B, R, H, D).transpose(1, 2)
    k = c[:, :, H * D:2 * H * D].view(B, S_kv, H, D).transpose(1, 2)
    v = c[:, :, 2 * H * D:].view(B, S_kv, H, D).transpose(1, 2)
    s = q @ k.transpose(-1, -2) / math.sqrt(D)
    mask = torch.arange(S_kv)[None, :] > (S_kv - R + torch.arange(R))[:, None]
    s = s.masked_fill(mask, float("-inf"))
    want = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B * R, H * D)
    assert fro_rel(o, want) < 5e-3


def test_prefix_cached_sampler_matches_the_whole_forward_sampler(dev, model):
    """predict_action_diff with the cached prefix (one prefill + 8 passes over the 1 + T suffix rows, mla_amd/infer.py) against the
    reference's control flow (8 whole forwards, reuse_prefix=False) on the same noise and FPS start indices: same function, other
    summation order -> the two chunks agree far inside the bound either has against the reference golden; the epsilon of one call agrees
    with the eval forward's; the captured HIP graph replays bit-identically to eager launches."""
    from mla_amd import infer
    m, gold = model
    ids, image, pc, proprio, noise, _ = infer_inputs()
    kw = dict(image=image[0], pointcloud=pc[0].numpy(), cur_robot_state=proprio[0, 0].numpy(), input_ids=ids, noise=noise, num_ddim_steps=8)
    full = m.predict_action_diff(reuse_prefix=False, **kw)
    cached = m.predict_action_diff(reuse_prefix=True, **kw)
    ref = gold["mla_ddim8_actions"][0]
    e_full, e_cached = np.linalg.norm(full - ref) / np.linalg.norm(ref), np.linalg.norm(cached - ref) / np.linalg.norm(ref)
    d = np.linalg.norm(cached - full) / np.linalg.norm(full)
    print(f"8-step DDIM chunk vs reference golden: whole forwards {e_full:.3e}, cached prefix {e_cached:.3e}; cached vs whole {d:.3e}")
    assert e_full < 1e-1 and e_cached < 1e-1 and d < 3e-2
    # one epsilon call
    with torch.inference_mode():
        _, eps_full = m.vlm(noise.to(dev), torch.tensor([91], device=dev), input_ids=ids.to(dev), images=image.to(dev), point_cloud=pc.to(dev),
                            proprio=proprio.to(dev), camera_name="rlbench_front")
    eps_model = infer.PrefixCachedEps.for_inputs(m.vlm, n_action_rows=4, input_ids=ids.to(dev), images=image.to(dev), point_cloud=pc.to(dev),
                                                 proprio=proprio.to(dev), camera_name="rlbench_front")
    _, eps_c = eps_model(noise.to(dev), torch.tensor([91], device=dev))
    assert eps_model.graph is not None, "the suffix pass was not captured into a graph"
    assert fro_rel(eps_c, eps_full.float().cpu()) < 2e-2
    assert fro_rel(eps_c, torch.from_numpy(gold["mla_eps_t91"])) < 5e-2
    _, eps_c2 = eps_model(noise.to(dev), torch.tensor([91], device=dev))        # replay
    old = infer._USE_GRAPH
    try:
        infer._USE_GRAPH = False
        _, eps_e = eps_model(noise.to(dev), torch.tensor([91], device=dev))     # eager launches on the same cache
    finally:
        infer._USE_GRAPH = old
    assert torch.equal(eps_c, eps_c2) and torch.equal(eps_c, eps_e)
    # the engine (cache buffers + captured graph) is reused for the next observation of the same shape: a new prefill, the same graph
    g0 = eps_model.graph
    image2 = image.clone()
    image2[:, :3] *= 0.5                                                        # (the mask channel stays all ones)
    again = infer.PrefixCachedEps.for_inputs(m.vlm, n_action_rows=4, input_ids=ids.to(dev), images=image2.to(dev), point_cloud=pc.to(dev),
                                             proprio=proprio.to(dev), camera_name="rlbench_front")
    assert again is eps_model and again.graph is g0
    _, eps_other = again(noise.to(dev), torch.tensor([91], device=dev))
    assert not torch.equal(eps_other, eps_c)                                    # another image -> another prefix -> another epsilon
    back = infer.PrefixCachedEps.for_inputs(m.vlm, n_action_rows=4, input_ids=ids.to(dev), images=image.to(dev), point_cloud=pc.to(dev),
                                            proprio=proprio.to(dev), camera_name="rlbench_front")
    assert torch.equal(back(noise.to(dev), torch.tensor([91], device=dev))[1], eps_c)
