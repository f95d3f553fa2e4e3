"""Import the REAL reference (/root/reference) inside the build container to pin the oracle.

Test infrastructure, build-container only: the reference never travels to the GPU box; what travels are the vectors this
produces (tests/golden/*.npz, written by oracle/capture_golden.py). Nothing here edits or copies reference source.
"""
import importlib.metadata as md
import os
import sys
from importlib.machinery import ModuleSpec
from unittest.mock import MagicMock

REF = os.environ.get("MLA_REFERENCE", "/root/reference")
SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_shim")
_MOCK = {"torchvision", "ipdb", "torch_geometric", "torch_scatter", "dlimp", "tensorflow", "tensorflow_datasets",
         "tensorflow_graphics", "absl", "wandb", "jsonlines", "peft", "termcolor", "open3d"}
_done = False


class _Finder:
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in _MOCK:
            return ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, m):
        pass


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "models", "mla"))


def setup():
    """Make `import models`, `import vla`, vendored `transformers` resolve to the reference. Idempotent."""
    global _done
    if _done:
        return
    if not available():
        raise RuntimeError(f"reference not found at {REF}")
    sys.dont_write_bytecode = True
    _v = md.version
    md.version = lambda n: {"tokenizers": "0.19.1", "huggingface-hub": "0.23.0", "huggingface_hub": "0.23.0"}.get(n) or _v(n)
    sys.meta_path.insert(0, _Finder())
    sys.path[:0] = [REF, SHIM]
    import models  # noqa: F401  (must be first: circular import modeling_llama <-> llama2)
    import models.vlm.prismatic as P
    P.visualize_generation_simple = lambda *a, **k: None
    _done = True


def build_reference_backbone(cfg_kwargs: dict):
    """The reference's LLMBackbone interface around a tiny real LlamaForCausalLM (eager attention, fake tokenizer)."""
    setup()
    import torch
    from models.backbones.llm.base_llm import LLMBackbone
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer

    class _Tok:
        vocab_size, pad_token_id, padding_side = cfg_kwargs["vocab_size"], cfg_kwargs["vocab_size"] - 1, "right"

        def encode(self, s, add_special_tokens=False):
            return [3]

    class _Backbone(LLMBackbone):
        def __init__(self):
            super().__init__("tiny-llama")
            self.llm = LlamaForCausalLM(LlamaConfig(**cfg_kwargs, attn_implementation="eager"))
            self.tokenizer = _Tok()

        def get_fsdp_wrapping_policy(self): return None
        def enable_gradient_checkpointing(self): pass
        def embed_input_ids(self, input_ids): return self.llm.get_input_embeddings()(input_ids)
        @property
        def prompt_builder_fn(self): return None
        @property
        def transformer_layer_cls(self): return LlamaDecoderLayer
        @property
        def half_precision_dtype(self): return torch.bfloat16
        @property
        def last_layer_finetune_modules(self): return ()

        def forward(self, **kw):
            return self.llm(**kw)

    return _Backbone()


def build_reference_mla(cfg_kwargs: dict, token_size: int, use_pointcloud=True, use_contrastive=True, future_action_window_size=0,
                        generation=None, use_tactile=False):
    """Tiny reference MLA: real PrismaticVLM/MLA/LlamaForCausalLM classes, eager attention, fake tokenizer."""
    setup()
    from models.mla import MLA
    from models.vlm.prismatic import PrismaticVLM
    bb = build_reference_backbone(cfg_kwargs)
    gen = generation or dict(use_generation=False)
    if gen.get("use_generation"):
        # the shipped constructor reads self.tactile_dim (prismatic.py:267) which only exists when use_tactile=True (:234-236), so
        # USE_GEN=true with USE_TAC=false (scripts/post_rlbench.sh) raises AttributeError; a class-level default lets it build
        PrismaticVLM.tactile_dim = 12
    vlm = PrismaticVLM("tiny", bb, token_size=token_size, action_dim=7, use_diff=True, use_pointcloud=use_pointcloud,
                       use_contrastive=use_contrastive, use_tactile=use_tactile, future_action_window_size=future_action_window_size, **gen)
    flags = {k: gen[k] for k in ("use_generation", "gen_image", "use_roi", "gen_pointcloud", "gen_tactile") if k in gen}
    mla = MLA(vlm, None, token_size=token_size, action_dim=7, future_action_window_size=future_action_window_size, use_diff=True,
              use_pointcloud=use_pointcloud, use_contrastive=use_contrastive, use_tactile=use_tactile, **flags)
    return mla
