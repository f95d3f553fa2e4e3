"""a checkpoint written by mla_amd's FSDPStrategy.save_checkpoint is read by the REFERENCE's own loader.

Build container only (imports /root/reference). Steps:
  1. build mla_amd's tiny MLA (post-training flags: every module key exists), load the recipe weights, shard it with FSDPStrategy on the
     CPU (torch LocalOps from tests/), write `step-...pt` with save_checkpoint (training/strategies/fsdp.py:100-141 layout);
  2. hand that file to the reference's `MLA.from_pretrained` (models/mla/model_mla.py:311-492) with the reference's own PrismaticVLM /
     LlamaForCausalLM classes -- every `load_state_dict(...)` in there is strict except the LLM's;
  3. compare the loaded reference model's state dict with the recipe weights BIT-EXACTLY, key by key;
  4. record the manifest (module key -> leaf -> shape, dtype) under tests/golden/checkpoint_manifest.json; the CPU test
     tests/test_host_logic.py::test_checkpoint_manifest_matches_reference_loader replays it without the reference.
"""
import json
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import recipe, ref_import  # noqa: E402

FLAGS = dict(use_generation=True, gen_image=True, use_roi=False, gen_pointcloud=True, gen_tactile=False)


def write_ours(tmp):
    from test_fsdp_gloo import TorchLocalOps
    from mla_amd.backbones import LLaMa2LLMBackbone
    from mla_amd.llama import LlamaConfig
    from mla_amd.mla import MLA
    from mla_amd.prismatic import PrismaticVLM
    from mla_amd.strategy import FSDPStrategy
    bb = LLaMa2LLMBackbone(config=LlamaConfig(**recipe.TINY_LLAMA), pad_to_multiple_of=1)
    vlm = PrismaticVLM("tiny", bb, token_size=recipe.TOKEN_SIZE, use_diff=True, use_pointcloud=True, use_contrastive=True, **FLAGS,
                       **recipe.GEN_TINY)
    m = MLA(vlm, None, token_size=recipe.TOKEN_SIZE, future_action_window_size=0, use_diff=True, use_pointcloud=True,
            use_contrastive=True, **FLAGS)
    want = {k: recipe.det_weight(k, v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(want, strict=True)
    m.freeze_backbones("post-training")
    strat = FSDPStrategy(m, "cpu", stage="post-training", local_ops=TorchLocalOps(), enable_gradient_checkpointing=False)
    strat.run_setup(100)
    return strat.save_checkpoint(tmp, global_step=7, epoch=1, train_loss=1.0, only_trainable=False), want


def main():
    with tempfile.TemporaryDirectory() as tmp:
        path, want = write_ours(tmp)
        ref_import.setup()
        from models.mla import MLA as RefMLA
        from models.vlm.prismatic import PrismaticVLM as RefVLM
        RefVLM.tactile_dim = 12          # the shipped constructor bug with USE_GEN and no tactile
        cfg = recipe.TINY_LLAMA | {"vocab_size": recipe.TINY_LLAMA["vocab_size"] + 1}
        ref = RefMLA.from_pretrained(None, path, "tiny", ref_import.build_reference_backbone(cfg), freeze_weights=False, action_dim=7,
                                     future_action_window_size=0, use_diff=True, use_pointcloud=True, use_contrastive=True,
                                     token_size=recipe.TOKEN_SIZE, **FLAGS, **recipe.GEN_TINY)
        got = ref.state_dict()
        ck = torch.load(path, map_location="cpu")["model"]
    missing = sorted(set(want) - set(got))
    extra = sorted(set(got) - set(want))
    assert not missing and not extra, (missing[:5], extra[:5])
    bad = [k for k in want if not (got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]))]
    assert not bad, f"{len(bad)} tensors differ after the reference loaded our checkpoint: {bad[:5]}"
    manifest = {mk: {leaf: [list(v.shape), str(v.dtype)] for leaf, v in sd.items()} for mk, sd in ck.items()}
    out = os.path.join(ROOT, "tests", "golden", "checkpoint_manifest.json")
    with open(out, "w") as fh:
        json.dump({"note": "module key -> leaf -> [shape, dtype] of a checkpoint written by mla_amd and loaded bit-exactly by the "
                           "reference's MLA.from_pretrained (oracle/crossload_checkpoint.py)",
                   "n_tensors": len(want), "manifest": manifest}, fh, indent=0, sort_keys=True)
    print(f"reference MLA.from_pretrained loaded {len(want)} tensors from our checkpoint bit-exactly; manifest -> {out}")


if __name__ == "__main__":
    main()
