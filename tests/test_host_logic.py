"""CPU: host-side logic of the drop-in modules (no kernels): action tokenizer (bit-exact), diffusion tables, sequence
splice plan vs the reference's per-sample loop (oracle restatement), module surface / state-dict names, LR schedule."""
import math
import os
import sys

import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import torch_oracle as O
from tests_shapes import MLA_TINY_SHAPES

G = os.path.join(os.path.dirname(__file__), "golden")


def test_action_tokenizer_bit_exact_vs_reference_golden():
    from mla_amd.action_tokenizer import ActionTokenizer
    from mla_amd.backbones import SyntheticLlamaTokenizer
    comp = np.load(os.path.join(G, "components.npz"))
    at = ActionTokenizer(SyntheticLlamaTokenizer(32000))
    ids = at.encode_ids(comp["at_actions"])
    assert ids.dtype == comp["at_ids"].dtype and np.array_equal(ids, comp["at_ids"])
    assert np.array_equal(at.decode_token_ids_to_actions(ids), comp["at_decoded"])
    assert ids.min() == 31744 and ids.max() == 31999 and at.action_token_begin_idx == 32000 - 257
    assert np.array_equal(at.encode_ids(np.zeros((0,))), np.zeros((0,), dtype=ids.dtype))            # empty input
    assert at.encode_ids(np.array([np.nextafter(1.0, 2.0), -np.inf, np.inf])).tolist() == [31744, 31999, 31744]


def test_diffusion_tables_match_reference_golden():
    from mla_amd.diffusion import create_diffusion
    comp = np.load(os.path.join(G, "components.npz"))
    d = create_diffusion(timestep_respacing="", noise_schedule="squaredcos_cap_v2", diffusion_steps=100)
    assert d.num_timesteps == 100
    assert np.array_equal(d.betas, comp["betas"])
    assert np.array_equal(d.sqrt_alphas_cumprod, comp["sqrt_ac"])
    assert np.array_equal(d.sqrt_one_minus_alphas_cumprod, comp["sqrt_1mac"])


@pytest.mark.parametrize("L,lens,T", [(16, [16, 13], 1), (12, [12, 12, 9, 5], 4), (8, [8], 1)])
def test_splice_plan_matches_reference_loop(L, lens, T):
    from mla_amd.prismatic import build_splice_plan
    B, nf, ins = len(lens), 513, 2 + T
    g = torch.Generator().manual_seed(L)
    ids = torch.randint(3, 500, (B, L), generator=g)
    ids[:, 0] = 1
    for b, n in enumerate(lens):
        ids[b, n - 1] = 2
        ids[b, n:] = 512
    am = ids != 512
    labels = torch.where(am, ids, torch.full_like(ids, -100))
    flat, k, mask, labs = build_splice_plan(ids, am, labels, nf, ins, 2)
    ks, ref_mask, ref_labs = O.splice_sequence(ids, am, labels, nf, T)
    assert torch.equal(k.squeeze(1), ks) and torch.equal(mask, ref_mask.bool()) and torch.equal(labs, ref_labs)
    # gather plan reproduces the reference's concatenation order on a pool of distinct integers
    S = L + nf + ins
    pool = torch.arange(B * S).view(B, S)          # [text L | fused nf | inserted ins]
    got = pool.reshape(-1)[flat].view(B, S)
    for b in range(B):
        kk = int(ks[b])
        z = torch.cat([pool[b, :1], pool[b, L:L + nf], pool[b, 1:L]])
        exp = torch.cat([z[:kk], pool[b, L + nf:], z[kk:]])
        assert torch.equal(got[b], exp)
    assert len(set(flat.tolist())) == flat.numel()  # injective -> the backward scatter needs no atomics


def test_module_surface_and_state_dict_names():
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig, LlamaDecoderLayer
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    bb = LLaMa2LLMBackbone(config=LlamaConfig(**recipe.TINY_LLAMA), pad_to_multiple_of=1)
    vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=True, use_contrastive=True,
                       use_generation=False)
    m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=True,
            use_contrastive=True)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == MLA_TINY_SHAPES      # == the reference's key set
    assert all(p.dtype == torch.float32 for p in m.parameters())                          # fp32 at load (scripts/train.py:303-307)
    assert bb.transformer_layer_cls is LlamaDecoderLayer and len(bb.tokenizer) == 513
    assert float(m.vlm.final_layer.mlp.fc2.weight.abs().max()) == 0.0                     # zero-init read-out (prismatic.py:320-321)
    m.freeze_backbones("finetune")
    assert m.trainable_module_keys == ["vlm.llm_backbone", "vlm.projector_2d", "vlm.proprio_embedder", "vlm.x_embedder",
                                       "vlm.t_embedder", "vlm.final_layer", "vlm.projector_3d"]
    assert not any(p.requires_grad for p in m.vlm.vision_tower_2d.parameters())
    assert not any(p.requires_grad for p in m.vlm.vision_tower_3d.parameters())
    pol = m.get_fsdp_wrapping_policy()
    units = [n for n, mod in m.named_modules() if pol(mod)]
    assert sum("layers." in u for u in units) == 9 and "vlm.vision_tower_2d" in units and "vlm.projector_3d" in units
    with pytest.raises(ValueError):
        m.freeze_backbones("align")
    # the reference's defaults (use_generation=True, gen_pointcloud=True, gen_tactile=True, prismatic.py:167-177) build the generation
    # manager with the point and tactile heads; use_tactile needs the point cloud
    full = PrismaticVLM("x", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, pointcloud_trans_dim=64, pointcloud_num_groups=4)
    assert full.generation_manager.get_module_keys() == ["pointcloud_gen_module", "tactile_gen_module"]
    assert "generation_manager" in full.all_module_keys and "tactile_embedder" not in full.all_module_keys
    with pytest.raises(ValueError):
        PrismaticVLM("x", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_generation=False, use_tactile=True, use_pointcloud=False)


def test_lr_schedule_and_camera_constants():
    from mla_amd.fuser import get_camera_params, projection_constants
    from mla_amd.strategy import cosine_with_warmup
    assert cosine_with_warmup(0, 10, 100) == 0.0 and cosine_with_warmup(10, 10, 100) == 1.0
    assert abs(cosine_with_warmup(55, 10, 100) - 0.5 * (1 + math.cos(math.pi * 0.5))) < 1e-12
    with pytest.raises(ValueError):
        get_camera_params("nope")
    Rw, tw, Ks = projection_constants("rlbench_front")
    assert abs(float(Ks[0, 0]) - (-307.7174807 * 3)) < 1e-3 and abs(float(Ks[0, 2]) - 336.0) < 1e-4


def test_generation_manager_state_dict_matches_reference():
    """Post-training heads: parameter / buffer names and shapes equal the reference's (captured in generation.npz)."""
    import os
    import numpy as np
    from mla_amd.generation import MultimodalGenerationManager
    from oracle import recipe
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "generation.npz"), allow_pickle=True)
    g = recipe.GEN_TINY
    mgr = MultimodalGenerationManager(
        token_size=recipe.TOKEN_SIZE, use_image_generation=True, num_image_gen_queries=g["num_image_gen_queries"],
        image_decoder_layers=g["image_decoder_layers"], image_decoder_heads=g["image_decoder_heads"], image_patch_size=42, use_roi=False,
        use_pointcloud_generation=True, pointcloud_trans_dim=g["pointcloud_trans_dim"], pointcloud_decoder_layers=g["pointcloud_decoder_layers"],
        pointcloud_decoder_heads=g["pointcloud_decoder_heads"], pointcloud_group_size=g["pointcloud_group_size"],
        pointcloud_num_groups=g["pointcloud_num_groups"])
    mine = {k: str(tuple(v.shape)) for k, v in mgr.state_dict().items()}
    ref = {str(n): str(s) for n, s in zip(gold["param_names"], gold["param_shapes"])}
    assert mine == ref


def test_post_training_model_keys_and_freeze():
    """BASELINE config[3] wiring on the host: state-dict keys of the whole model, trainable set, module keys."""
    import os
    import numpy as np
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    from oracle import recipe
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "mla_tiny_e2e_gen.npz"), allow_pickle=True)
    bb = LLaMa2LLMBackbone(config=LlamaConfig(**recipe.TINY_LLAMA), pad_to_multiple_of=1)
    flags = dict(use_generation=True, gen_image=True, use_roi=False, gen_pointcloud=True, gen_tactile=False)
    vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=True, use_contrastive=True, **flags,
                       **recipe.GEN_TINY)
    m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=True, use_contrastive=True,
            **flags)
    mine = {k: str(tuple(v.shape)) for k, v in m.state_dict().items()}
    assert mine == {str(n): str(s) for n, s in zip(gold["param_names"], gold["param_shapes"])}
    m.freeze_backbones("post-training")
    assert "vlm.generation_manager" in m.trainable_module_keys and "vlm.generation_manager" in m.all_module_keys
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    assert set(str(n) for n in gold["grad_names"]) <= trainable
    assert not any(n.startswith("vlm.vision_tower_2d") or n.startswith("vlm.vision_tower_3d") for n in trainable)
    roi_vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_generation=True, gen_image=True, use_roi=True,
                           gen_pointcloud=False, gen_tactile=False, **recipe.GEN_TINY)
    assert roi_vlm.generation_manager.image_gen_module.use_roi and roi_vlm.use_roi


def test_hf_llama_weight_files_load_into_backbone(tmp_path):
    """HF-named safetensors shards (32000-row tables) load into the resized (32064-row style) backbone."""
    from safetensors.torch import save_file
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    cfg = dict(vocab_size=500, hidden_size=128, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1, rms_norm_eps=1e-5)
    bb = LLaMa2LLMBackbone(config=LlamaConfig(**cfg), pad_to_multiple_of=64)
    own = bb.llm.state_dict()
    g = torch.Generator().manual_seed(0)
    hf = {}
    for k, v in own.items():
        if "contrastive" in k:
            continue
        shape = (500, 128) if k in ("model.embed_tokens.weight", "lm_head.weight") else tuple(v.shape)
        hf[k] = torch.randn(*shape, generator=g)
    hf["model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.ones(64)
    keys = sorted(hf)
    save_file({k: hf[k] for k in keys[: len(keys) // 2]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({k: hf[k] for k in keys[len(keys) // 2:]}, str(tmp_path / "model-00002-of-00002.safetensors"))
    before_tail = own["model.embed_tokens.weight"][500:].clone()
    info = bb.load_hf_checkpoint(str(tmp_path))
    assert info["loaded"] == len(hf) - 1 and info["skipped"] == ["model.layers.0.self_attn.rotary_emb.inv_freq"]
    now = bb.llm.state_dict()
    assert now["model.embed_tokens.weight"].shape[0] == 512
    assert torch.equal(now["model.embed_tokens.weight"][:500], hf["model.embed_tokens.weight"])
    assert torch.equal(now["model.embed_tokens.weight"][500:], before_tail)
    assert torch.equal(now["model.layers.1.mlp.down_proj.weight"], hf["model.layers.1.mlp.down_proj.weight"])


def test_checkpoint_layout_roundtrip(tmp_path):
    """save_checkpoint writes the reference's layout ({"model": {module_key: {leaf: fp32}}}, `vlm.` dropped, file name pattern,
    fsdp.py:100-141); MLA.from_pretrained reads it back with the reference's per-module rules (model_mla.py:360-465)."""
    import numpy as np
    from test_fsdp_gloo import TorchLocalOps
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    from mla_amd.strategy import FSDPStrategy
    from oracle import recipe
    flags = dict(use_generation=True, gen_image=True, use_roi=False, gen_pointcloud=True, gen_tactile=False)

    def backbone():
        return LLaMa2LLMBackbone(config=LlamaConfig(**recipe.TINY_LLAMA), pad_to_multiple_of=1)
    vlm = PrismaticVLM("tiny", backbone(), token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=True, use_contrastive=True, **flags,
                       **recipe.GEN_TINY)
    m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=True, use_contrastive=True,
            **flags)
    want = {k: recipe.det_weight(k, v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(want, strict=True)
    m.freeze_backbones("post-training")
    strat = FSDPStrategy(m, "cpu", stage="post-training", local_ops=TorchLocalOps(), enable_gradient_checkpointing=False)
    strat.run_setup(100)
    assert next(m.parameters()).dtype == torch.bfloat16            # compute weights; fp32 masters live in the shards
    path = strat.save_checkpoint(tmp_path, global_step=12, epoch=3, train_loss=0.123456)
    assert path.name == "step-000012-epoch-03-loss=0.1235.pt" and path.parent.name == "checkpoints"
    ck = torch.load(path, map_location="cpu")
    assert list(ck) == ["model"]
    assert set(ck["model"]) == {"vision_tower_2d", "projector_2d", "llm_backbone", "proprio_embedder", "x_embedder", "t_embedder",
                                "final_layer", "vision_tower_3d", "projector_3d", "generation_manager"}
    assert "llm.model.layers.0.self_attn.q_proj.weight" in ck["model"]["llm_backbone"]
    assert "image_gen_module.intent_decoder.layers.0.self_attn.in_proj_weight" in ck["model"]["generation_manager"]
    for mkey, sd in ck["model"].items():
        for leaf, v in sd.items():
            ref = want[f"vlm.{mkey}.{leaf}"]
            assert v.dtype == ref.dtype and torch.equal(v, ref), (mkey, leaf)       # fp32 masters, bit-exact
    assert strat.save_checkpoint(tmp_path, 13, 3).name == "step-000013-epoch-03-loss=inf.pt"
    m2 = MLA.from_pretrained(None, path, "tiny", backbone(), freeze_weights=False, action_dim=7, future_action_window_size=0,
                             use_diff=True, use_pointcloud=True, use_contrastive=True, **flags, **recipe.GEN_TINY)
    got = m2.state_dict()
    assert all(torch.equal(got[k], want[k]) for k in want)
    assert "generation_manager.image_gen_module" in m2.loaded_module_keys and "llm_backbone" in m2.loaded_module_keys
    # a checkpoint whose embedders were trained for another action width keeps the fresh initialisation (model_mla.py:400, 417)
    ck["model"]["proprio_embedder"]["mlp.fc1.weight"] = torch.zeros(recipe.TOKEN_SIZE, 14)
    ck["model"].pop("projector_3d")
    torch.save(ck, tmp_path / "other.pt")
    m3 = MLA.from_pretrained(None, tmp_path / "other.pt", "tiny", backbone(), action_dim=7, future_action_window_size=0, use_diff=True,
                             use_pointcloud=True, use_contrastive=True, use_generation=False)
    assert "proprio_embedder" not in m3.loaded_module_keys and "projector_3d" not in m3.loaded_module_keys
    assert not any(p.requires_grad for p in m3.parameters()) and not m3.vlm.training


def test_checkpoint_manifest_matches_reference_loader(tmp_path):
    """oracle/crossload_checkpoint.py wrote a checkpoint with FSDPStrategy.save_checkpoint, loaded it with the
    REFERENCE's MLA.from_pretrained (models/mla/model_mla.py:311-492; strict load_state_dict per module) and compared all 323 tensors
    bit-exactly; it recorded the file's manifest. Here (no reference needed) the same writer must still produce exactly that manifest:
    module keys, leaf names in order, shapes and dtypes."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from oracle import crossload_checkpoint as cc
    path, want = cc.write_ours(tmp_path)
    ck = torch.load(path, map_location="cpu")["model"]
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "checkpoint_manifest.json")))
    assert gold["n_tensors"] == len(want) == sum(len(v) for v in ck.values())
    assert sorted(ck) == sorted(gold["manifest"])
    for mk, sd in ck.items():
        assert sorted(sd.keys()) == sorted(gold["manifest"][mk].keys()), mk
        for leaf, v in sd.items():
            assert [list(v.shape), str(v.dtype)] == gold["manifest"][mk][leaf], (mk, leaf)
            assert torch.equal(v, want[f"vlm.{mk}.{leaf}"]), (mk, leaf)


def test_action_unnormalisation_rules():
    """model_mla.py:667-704: proprio q01/q99 normalisation + clip; action clip, gripper binarisation at 0.5, masked affine map."""
    import numpy as np
    from mla_amd.mla import MLA
    m = MLA.__new__(MLA)
    m.norm_stats = {"a": {"action": {"q01": [0.0] * 7, "q99": [2.0] * 7, "mask": [True] * 6 + [False]},
                          "proprio": {"q01": [-1.0] * 7, "q99": [3.0] * 7}}}
    x = np.array([[-2.0, -1.0, 0.0, 0.5, 1.0, 3.0, 0.4], [0.1] * 6 + [0.6]])
    got = m.unnormalize_actions(x.copy())
    assert np.allclose(got[0], [0.0, 0.0, 1.0, 1.5, 2.0, 2.0, 0.0]) and got[1, 6] == 1.0
    assert np.allclose(m.normalize_proprio(np.array([-1.0, 3.0, 1.0, 5.0, -9.0, 0.0, 2.0])), [-1, 1, 0, 1, -1, -0.5, 0.5], atol=1e-6)
    m.norm_stats["b"] = m.norm_stats["a"]
    import pytest
    with pytest.raises(AssertionError):
        m.get_action_stats(None)
    assert m.get_action_dim("b") == 7


def test_padded_collator_matches_reference_semantics():
    """PaddedCollatorForActionPrediction (util/data_utils.py:88-196): padding / truncation / mask / optional fields; cross-checked against
    the reference's own class when it is importable (build container)."""
    from mla_amd.data_utils import IGNORE_INDEX, PaddedCollatorForActionPrediction
    g = torch.Generator().manual_seed(0)

    def inst(n, tactile):
        d = dict(input_ids=torch.randint(3, 100, (n,), generator=g), labels=torch.randint(3, 100, (n,), generator=g),
                 images={"front_image": torch.randn(4, 8, 8, generator=g)}, next_images={"front_image": torch.randn(3, 8, 8, generator=g)},
                 point_cloud=torch.randn(16, 3, generator=g), next_point_cloud=torch.randn(16, 3, generator=g),
                 actions=torch.randn(1, 7, generator=g), action_masks=torch.ones(1, dtype=torch.bool), proprio=torch.randn(1, 7, generator=g),
                 dataset_name="rlbench")
        if tactile:
            d.update(tactile=torch.randn(12, generator=g), gripper_xyz=torch.randn(3, generator=g), next_tactile=torch.randn(12, generator=g))
        return d
    for tactile in (False, True):
        batch = [inst(9, tactile), inst(14, tactile), inst(5, tactile)]
        out = PaddedCollatorForActionPrediction(model_max_length=12, pad_token_id=512)(batch)
        assert out["input_ids"].shape == (3, 12) and out["labels"].shape == (3, 12)
        assert torch.equal(out["input_ids"][0, :9], batch[0]["input_ids"]) and bool((out["input_ids"][0, 9:] == 512).all())
        assert torch.equal(out["input_ids"][1], batch[1]["input_ids"][:12])                       # truncated
        assert bool((out["labels"][2, 5:] == IGNORE_INDEX).all()) and torch.equal(out["attention_mask"], out["input_ids"] != 512)
        assert out["images"]["front_image"].shape == (3, 4, 8, 8) and out["point_cloud"].shape == (3, 16, 3)
        assert (out["tactile"] is None) == (not tactile) and (out["gripper_xyz"] is None) == (not tactile)
        assert out["dataset_names"] == ["rlbench"] * 3
        try:
            from oracle import ref_import
            if not ref_import.available():
                raise ImportError
            ref_import.setup()
            from util.data_utils import PaddedCollatorForActionPrediction as Ref
        except Exception:
            continue
        ref = Ref(model_max_length=12, pad_token_id=512)(batch)
        assert set(ref) == set(out)
        for k, v in ref.items():
            if torch.is_tensor(v):
                assert torch.equal(v, out[k]), k
            elif isinstance(v, dict):
                assert all(torch.equal(v[kk], out[k][kk]) for kk in v), k
            else:
                assert v == out[k], k


def test_rlds_batch_transform_matches_reference_golden():
    """Data-side transform: prompt text, action-token order, label masking, mask channel, tactile clean-up, against
    vectors captured from the reference's own classes with the same toy tokenizer / image transform (oracle/capture_golden_transform.py)."""
    from mla_amd.data_utils import ActionTokenizer, PurePromptBuilder, RLDSBatchTransform
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "transform.npz"), allow_pickle=True)
    tok = recipe.ToyTokenizer()
    for name, kw in recipe.RLDS_CASES.items():
        at = None if name == "no_action_tok" else ActionTokenizer(tok)
        tf = RLDSBatchTransform(at, tok, recipe.ToyImageTransform(), PurePromptBuilder, predict_stop_token=name != "no_stop",
                                use_pointcloud=kw.get("with_pc", False), use_tactile=kw.get("with_tactile", False))
        out = tf(recipe.make_rlds_sample(**kw))
        seen = 0
        for k, v in out.items():
            if k == "images":
                assert sorted(f"{name}::images.{c}" for c in v) == sorted(f for f in gold.files if f.startswith(f"{name}::images."))
                for cam, t in v.items():
                    assert t.shape == (4, 672, 672) and np.array_equal(t[:, ::37, ::41].numpy(), gold[f"{name}::images.{cam}"]), (name, cam)
                    seen += 1
            elif torch.is_tensor(v):
                got = (v[:, ::37, ::41] if k == "next_images" else v).numpy()
                ref = gold[f"{name}::{k}"]
                assert got.dtype == ref.dtype and np.array_equal(got, ref), (name, k)
                seen += 1
            elif v is None:
                assert str(gold[f"{name}::{k}"]) == "None", (name, k)
                seen += 1
        assert seen == sum(f.startswith(name + "::") for f in gold.files)
        assert out["dataset_name"] == b"rlbench"
    at = ActionTokenizer(tok)
    a = np.linspace(-1.5, 1.5, 1001)
    assert np.array_equal(at.encode_ids(a), gold["at_ids"]) and at(a[::100]) == str(gold["at_str"])
    assert np.array_equal(at.decode_token_ids_to_actions(np.arange(tok.vocab_size - 257, tok.vocab_size)), gold["at_decode"])
    assert at.action_token_begin_idx == int(gold["at_begin"]) and at.vocab_size == 256
    pb = PurePromptBuilder("openvla")
    pb.add_turn("human", " <image> What now? ")
    pb.add_turn("gpt", "")
    assert pb.get_prompt() == str(gold["pb_prompt"]) and pb.get_potential_prompt("and then?") == str(gold["pb_potential"])
    # the transform feeds the collator
    from mla_amd.data_utils import PaddedCollatorForActionPrediction
    tf = RLDSBatchTransform(ActionTokenizer(tok), tok, recipe.ToyImageTransform(), PurePromptBuilder, use_pointcloud=True, use_tactile=True)
    items = [tf(recipe.make_rlds_sample(with_tactile=True, with_pc=True, seed=s)) for s in (0, 1)]
    batch = PaddedCollatorForActionPrediction(64, tok.pad_token_id)(items)
    assert batch["input_ids"].shape == batch["labels"].shape and batch["images"]["front_image"].shape == (2, 4, 672, 672)
    assert batch["tactile"].shape == (2, 12) and batch["point_cloud"].shape == (2, 64, 3)


def test_prompt_builder_selection():
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.data_utils import PurePromptBuilder
    bb = LLaMa2LLMBackbone.__new__(LLaMa2LLMBackbone)
    bb.identifier = "llama2-7b-pure"
    assert bb.prompt_builder_fn is PurePromptBuilder
    bb.identifier = "llama2-7b-chat"
    with pytest.raises(ValueError):
        bb.prompt_builder_fn


def test_run_vla_training_and_get_train_strategy_surface(tmp_path):
    """The caller-facing surface scripts/train.py uses (training/materialize.py:22-68, base_strategy_mla.py:251-404): the factory's
    signature / registry / error, and the loop's control flow on a toy module with MLA's forward signature -- batches from an
    IterableDataset through the collator, the reference's metric commits, one optimizer step per accumulation window, lr / epoch /
    global_step bookkeeping, a checkpoint at max_steps, termination."""
    import inspect
    import torch.nn as nn
    from torch.utils.data import IterableDataset
    from test_fsdp_gloo import TorchLocalOps
    from mla_amd import strategy as S

    sig = inspect.signature(S.get_train_strategy)
    assert list(sig.parameters)[:13] == ["train_strategy", "vlm", "device_id", "stage", "epochs", "max_steps", "global_batch_size",
                                         "per_device_batch_size", "learning_rate", "weight_decay", "max_grad_norm", "lr_scheduler_type",
                                         "warmup_ratio"]
    assert set(S.TRAIN_STRATEGIES) == {"fsdp-shard-grad-op", "fsdp-full-shard"}
    with pytest.raises(ValueError, match="is not supported"):
        S.get_train_strategy("ddp", None, 0, "finetune", 1, None, 8, 8, 1e-3, 0.0, 1.0, "constant", 0.0)
    rv = inspect.signature(S.FSDPStrategy.run_vla_training)
    assert list(rv.parameters)[1:] == ["vla_dataset", "collator", "metrics", "save_interval", "save_full_model", "use_diff",
                                       "use_pointcloud", "use_tactile", "use_contrastive", "camera_name", "use_generation", "gen_image",
                                       "gen_pointcloud", "gen_tactile", "repeated_diffusion_steps"]

    class ToyVLA(nn.Module):
        all_module_keys = trainable_module_keys = ["body"]

        def __init__(self):
            super().__init__()
            self.body = nn.Linear(4, 1)
            self.llm_backbone = type("B", (), {"enable_gradient_checkpointing": lambda self: None})()
            self.seen = []

        def get_fsdp_wrapping_policy(self):
            return lambda m: False

        def forward(self, input_ids=None, actions=None, camera_name=None, point_cloud=None, next_images=None, repeated_diffusion_steps=None,
                    **kw):
            self.seen.append((camera_name, point_cloud is not None, next_images is not None, repeated_diffusion_steps, tuple(actions.shape)))
            loss = (self.body(actions.to(self.body.weight.dtype)) ** 2).mean()
            z = torch.zeros(())
            return {"total_loss": loss, "diff_loss": loss, "img_pc_contrastive_loss": z, "tactile_contrastive_loss": z,
                    "image_gen_loss": z, "point_cloud_gen_loss": z, "tactile_gen_loss": z}, None

    class DS(IterableDataset):
        def __iter__(self):
            i = 0
            while True:                              # the RLDS contract: infinite iterator, __len__ = dataset size
                g = torch.Generator().manual_seed(i)
                yield {"actions": torch.randn(4, generator=g), "i": i}
                i += 1

        def __len__(self):
            return 12

    def collate(items):
        z = torch.zeros(len(items), 1)
        return {"actions": torch.stack([it["actions"] for it in items]), "input_ids": z.long(), "attention_mask": z.bool(), "labels": z.long(),
                "images": {"front_image": z}, "next_images": {"front_image": z}, "proprio": z, "point_cloud": z}

    class Metrics:
        global_step, run_dir = 0, tmp_path

        def __init__(self):
            self.log, self.pushed = [], 0

        def get_status(self):
            return "status"

        def commit(self, **kw):
            self.log.append(kw)
            if "global_step" in kw:
                self.global_step = kw["global_step"]

        def push(self):
            self.pushed += 1
            return "status"

    vla = ToyVLA()
    strat = S.get_train_strategy("fsdp-full-shard", vla, "cpu", "finetune", epochs=1, max_steps=None, global_batch_size=4,
                                 per_device_batch_size=2, learning_rate=1e-2, weight_decay=0.0, max_grad_norm=1.0,
                                 lr_scheduler_type="constant", warmup_ratio=0.0, enable_gradient_checkpointing=False,
                                 reduce_in_full_precision=True)
    strat.local_ops = TorchLocalOps()
    assert strat.grad_accumulation_steps == 2 and strat.sharding_strategy == "full-shard"
    strat.run_setup(run_dir=tmp_path, n_train_examples=12)
    w0 = strat.sharded.full_state_dict_fp32()["body.weight"].clone()
    m = Metrics()
    strat.run_vla_training(DS(), collate, m, save_interval=1, use_diff=True, use_pointcloud=False, camera_name="rlbench_front",
                           use_generation=False, repeated_diffusion_steps=3)
    # 12 samples / per-device batch 2 = 6 batches per epoch, 2 per optimizer step -> 3 optimizer steps, then the loop returns
    assert m.global_step == 3 and strat.step == 3 and m.pushed == 3 and len(vla.seen) == 6
    assert all(s == ("rlbench_front", False, False, 3, (2, 4)) for s in vla.seen)      # flags gate the optional streams (:306-322)
    steps = [kw for kw in m.log if "global_step" in kw]
    assert [kw["global_step"] for kw in steps] == [1, 2, 3] and all(kw["lr"] == 1e-2 and kw["update_step_time"] for kw in steps)
    assert [kw["epoch"] for kw in steps] == [0, 0, 1]                                    # (global_step + 1) // (12 // 4)
    assert sum(1 for kw in m.log if "diff_loss" in kw) == 6
    assert not torch.equal(strat.sharded.full_state_dict_fp32()["body.weight"], w0)
    ck = sorted((tmp_path / "checkpoints").glob("*.pt"))
    assert len(ck) == 1 and ck[0].name.startswith("step-000003-epoch-01-loss=")          # end of epoch 1, save_interval 1
    with pytest.warns(UserWarning, match="Optimizer checkpoint not found"):
        strat.load_optimizer_and_scheduler(ck[0])


def test_collective_knobs_and_buffer_arenas():
    """Round 4 plumbing around the sharded path (no GPU, no process group): the environment knobs bench.py applies before creating the
    RCCL communicator, and the per-kind buffer arenas (2 MiB-aligned slices, exact capacity accounting)."""
    import torch
    from mla_amd.fsdp import _Arenas, apply_rccl_env
    env = {"MLA_RCCL_MAX_CHANNELS": "8", "MLA_GEMM_CUS": "240", "MLA_FSDP_INPLACE_RS": "0", "UNRELATED": "1"}
    done = apply_rccl_env(env)
    assert env["NCCL_MAX_NCHANNELS"] == "8" and done == {"NCCL_MAX_NCHANNELS": "8", "MLA_GEMM_CUS": "240", "MLA_FSDP_INPLACE_RS": "0"}
    assert apply_rccl_env({}) == {}
    ar = _Arenas({"master": 1000, "exp_avg": 24, "gshard": 0}, torch.device("cpu"), slices=3)
    assert ar.has("master") and ar.has("exp_avg") and not ar.has("gshard") and not ar.has("flat16")
    a, b, c = ar.take("master", 400), ar.take("master", 600), ar.take("master", 0)
    assert a.numel() == 400 and b.numel() == 600 and c.numel() == 0
    assert a.data_ptr() % 16 == 0 and (b.data_ptr() - a.data_ptr()) % _Arenas.ALIGN_BYTES == 0 and b.data_ptr() > a.data_ptr()
    a.fill_(1.0)
    b.fill_(2.0)
    assert float(a.sum()) == 400.0 and float(b.sum()) == 1200.0            # disjoint
    import pytest
    with pytest.raises(AssertionError):
        for _ in range(8):
            ar.take("master", 600)                                          # beyond the planned capacity: loud, never silent overlap


def test_shared_prefix_row_plan_matches_a_per_sample_loop():
    """mla_amd.prismatic.shared_prefix_plan (round 6: the shared-prefix layout for prompts of different lengths) against the layout
    written out sample by sample: [BOS | fused | text[1:k] | proprio] + R x [t | x (T rows) | text[k]] with copy r of sample i at
    r * B + i in the t / x pools, zero rows behind the sample's valid rows, positions = the row's position in the reference's TILED
    sequence (models/vlm/prismatic.py:981-1038 builds that sequence once per copy)."""
    from mla_amd.prismatic import shared_prefix_plan
    B, L, n_fused, T, R = 3, 12, 5, 2, 4
    k_all = torch.tensor([11, 7, 9])
    idx, pos, P_i, V_i, S, in_suffix, w = shared_prefix_plan(k_all, B, L, n_fused, T, R)
    s = T + 2
    assert P_i.tolist() == [k + n_fused + 1 for k in k_all.tolist()] and V_i.tolist() == [p + R * s for p in P_i.tolist()]
    assert S % 4 == 0 and S >= max(V_i.tolist()) and S - max(V_i.tolist()) < 4
    o_fus, o_pro = B * L, B * L + B * n_fused
    o_t = o_pro + B
    o_x = o_t + R * B
    o_zero = o_x + R * B * T
    for i in range(B):
        k = int(k_all[i])
        want = [i * L] + [o_fus + i * n_fused + f for f in range(n_fused)] + [i * L + c for c in range(1, k)] + [o_pro + i]
        wpos = list(range(len(want)))
        P = len(want)
        for r in range(R):
            want += [o_t + r * B + i] + [o_x + (r * B + i) * T + c for c in range(T)] + [i * L + k]
            wpos += [P + c for c in range(s)]
        assert idx[i, :len(want)].tolist() == want, i
        assert pos[i, :len(want)].tolist() == wpos, i
        assert (idx[i, len(want):] == o_zero).all()
        assert in_suffix[i].sum() == R * s and not in_suffix[i, :P].any()
    # equal prompts degenerate to the same prefix length for every sample
    idx2, pos2, P2, V2, S2, _, _ = shared_prefix_plan(torch.tensor([9, 9]), 2, 10, 5, 1, 4)
    assert P2.tolist() == [15, 15] and S2 == 28 and pos2[0, 15:27].tolist() == [15, 16, 17] * 4


def test_lazy_lm_output_materialises_once_and_adds_the_contrastive_terms_in_order():
    """mla_amd.modeling_outputs.CausalLMOutputWithPast (round 6): with `lazy_lm` the logits / loss are produced on first access, exactly
    once, and loss = CE + img_pc + tactile in the reference's order (modeling_llama.py:1272-1303); without it the fields behave like
    the plain dataclass of the reference (modeling_outputs.py:706-713)."""
    from mla_amd.modeling_outputs import CausalLMOutputWithPast
    calls = []

    def lm():
        calls.append(1)
        return torch.ones(2, 3), torch.tensor(2.0)
    out = CausalLMOutputWithPast(img_pc_contrastive_loss=torch.tensor(0.5), tactile_contrastive_loss=torch.tensor(0.25), lazy_lm=lm,
                                 hidden_states=(torch.zeros(1),))
    assert out.lm_head_pending and not calls and "<lazy>" in repr(out)
    assert float(out.loss) == 2.75 and len(calls) == 1 and not out.lm_head_pending
    assert out.logits.shape == (2, 3) and len(calls) == 1 and out["loss"] is out.loss and out[0] is out.loss
    plain = CausalLMOutputWithPast(loss=torch.tensor(1.0), logits=None, hidden_states=None)
    assert float(plain.loss) == 1.0 and plain.logits is None and not plain.lm_head_pending
    none = CausalLMOutputWithPast(lazy_lm=lambda: (torch.zeros(1), None), img_pc_contrastive_loss=torch.tensor(1.0))
    assert none.loss is None and none.logits is not None          # no labels -> no CE -> loss stays None (the reference would raise on None + tensor)


def test_readout_output_keeps_the_dense_last_hidden_state_lazy():
    """Read-out mode (round 6, opt-in MLA.readout_rows_only): `readout_hidden` holds the rows the caller asked for; the dense final hidden
    state is appended to `hidden_states` on FIRST access, once, and neither repr() nor the other fields trigger it; assigning
    `hidden_states` drops the pending computation."""
    from mla_amd.modeling_outputs import CausalLMOutputWithPast
    calls = []

    def last():
        calls.append(1)
        return torch.full((1,), 9.0)
    out = CausalLMOutputWithPast(hidden_states=(torch.zeros(1), torch.ones(1)), lazy_last=last, readout_hidden=torch.full((2, 4), 3.0),
                                 img_pc_contrastive_loss=torch.tensor(0.5))
    assert out.last_hidden_pending and not calls and "hidden_states=<lazy>" in repr(out) and not calls
    assert out.readout_hidden.shape == (2, 4) and float(out.img_pc_contrastive_loss) == 0.5 and not calls
    hs = out.hidden_states
    assert len(hs) == 3 and float(hs[-1]) == 9.0 and calls == [1] and not out.last_hidden_pending
    assert out.hidden_states is hs and calls == [1]
    other = CausalLMOutputWithPast(hidden_states=(torch.zeros(1),), lazy_last=last)
    other.hidden_states = None
    assert other.hidden_states is None and not other.last_hidden_pending and calls == [1]
    dense = CausalLMOutputWithPast(hidden_states=(torch.zeros(1),))
    assert not dense.last_hidden_pending and len(dense.hidden_states) == 1 and dense.readout_hidden is None
