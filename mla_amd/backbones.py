"""LLM backbone wrapper with the reference's surface (models/backbones/llm/base_llm.py:37-241, llama2.py:55-104).

The reference downloads meta-llama/Llama-2-7b-hf; there is no network here, so the backbone is built from a config with
random-init weights (checkpoint layout / loading is SURVEY 8f rank 1). A minimal tokenizer stand-in provides the few
attributes the model code reads (vocab size incl. <PAD>, pad id, single-id encode for the trigger strings).
"""
from __future__ import annotations

from typing import Callable, Optional, Type

import torch
import torch.nn as nn

from .llama import LlamaConfig, LlamaDecoderLayer, LlamaForCausalLM
from .modeling_outputs import CausalLMOutputWithPast


class SyntheticLlamaTokenizer:
    """Stand-in for LlamaTokenizerFast: ids only (no text). vocab 32000 + <PAD> (llama2.py:75-77)."""

    def __init__(self, vocab_size: int = 32000):
        self.base_vocab = vocab_size
        self.added = ["<PAD>"]
        self.pad_token_id = vocab_size
        self.bos_token_id, self.eos_token_id = 1, 2
        self.padding_side = "right"
        self.model_max_length = 2048

    def __len__(self):
        return self.base_vocab + len(self.added)

    @property
    def vocab_size(self):
        return self.base_vocab

    def add_special_tokens(self, d):
        toks = d.get("additional_special_tokens", []) + ([d["pad_token"]] if "pad_token" in d else [])
        new = [t for t in toks if t not in self.added]
        self.added += new
        return len(new)

    _warned = False

    def encode(self, s, add_special_tokens=False):
        if not SyntheticLlamaTokenizer._warned:
            SyntheticLlamaTokenizer._warned = True
            import warnings
            warnings.warn("SyntheticLlamaTokenizer.encode is a hash, not the Llama-2 vocabulary: pass `tokenizer=` or `tokenizer_path=` "
                          "to LLaMa2LLMBackbone for real prompts")
        return [3 + (sum(map(ord, s)) % 1000)]


class LLMBackbone(nn.Module):
    def __init__(self, llm_backbone_id: str) -> None:
        super().__init__()
        self.identifier = llm_backbone_id
        self.llm = None
        self.tokenizer = None

    def get_tokenizer(self):
        return self.tokenizer


class LLaMa2LLMBackbone(LLMBackbone):
    def __init__(self, llm_backbone_id: str = "llama2-7b-pure", llm_max_length: int = 2048, hf_token: Optional[str] = None,
                 inference_mode: bool = False, use_flash_attention_2: bool = True, llm_vision_layers: int = 1,
                 config: Optional[LlamaConfig] = None, pad_to_multiple_of: int = 64, tokenizer=None,
                 tokenizer_path: Optional[str] = None, **kwargs) -> None:
        """``tokenizer``: a ready tokenizer object (the reference's `AutoTokenizer.from_pretrained(hf_hub_path, ...)`,
        base_llm.py:138-150); ``tokenizer_path``: a local directory with the Llama-2 tokenizer files, loaded with the vendored
        `transformers.AutoTokenizer` (`model_max_length=llm_max_length, padding_side="right"`), after which `<PAD>` is added like
        llama2.py:75. With neither, a SYNTHETIC stand-in is installed (hash `encode`, 32000 + `<PAD>` ids): right for the benchmark
        and the tests, wrong for real prompts -- `string2idx`, the embedding resize and `predict_action_diff` prompts depend on the
        real vocabulary, so `get_tokenizer()` / `encode` warn once when the stand-in is used."""
        super().__init__(llm_backbone_id)
        self.llm_max_length, self.inference_mode = llm_max_length, inference_mode
        cfg = config or LlamaConfig()
        if tokenizer is None and tokenizer_path is not None:
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(tokenizer_path, model_max_length=llm_max_length, padding_side="right")
            tokenizer.add_special_tokens({"pad_token": "<PAD>"})                       # llama2.py:75
        self.tokenizer = tokenizer if tokenizer is not None else SyntheticLlamaTokenizer(cfg.vocab_size)
        self.llm = LlamaForCausalLM(cfg)
        # llama2.py:75-77: add <PAD>, resize embeddings padded to a multiple of 64 (32001 -> 32064)
        self.llm.resize_token_embeddings(len(self.tokenizer), pad_to_multiple_of=pad_to_multiple_of)
        self.llm.config.pad_token_id = self.tokenizer.pad_token_id
        self.llm.config.use_cache = False

    def load_hf_checkpoint(self, path: str) -> dict:
        """base_llm.py:120-136 loads `meta-llama/Llama-2-7b-hf` through HF `from_pretrained` BEFORE the tokenizer resize; here the
        HF weight files (``*.safetensors`` shards, else ``pytorch_model*.bin``) of a local directory are read directly -- HF's
        parameter names are this module's names. Embedding / lm_head tables with fewer rows than the resized ones (32000 vs
        32064) fill the leading rows, exactly what ``resize_token_embeddings`` keeps. Returns {"loaded": n, "skipped": [...]}."""
        import glob
        import os
        files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if files:
            from safetensors.torch import load_file
            shards = (load_file(f) for f in files)
        else:
            files = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
            if not files:
                raise FileNotFoundError(f"no *.safetensors or pytorch_model*.bin under {path}")
            shards = (torch.load(f, map_location="cpu") for f in files)
        own = self.llm.state_dict()
        loaded, skipped = 0, []
        with torch.no_grad():
            for shard in shards:
                for k, v in shard.items():
                    if k not in own:
                        skipped.append(k)                      # e.g. rotary inv_freq buffers of older exports
                        continue
                    dst = own[k]
                    if dst.shape == v.shape:
                        dst.copy_(v)
                    elif dst.dim() == 2 and v.dim() == 2 and dst.shape[1] == v.shape[1] and dst.shape[0] > v.shape[0]:
                        dst[:v.shape[0]].copy_(v)
                    else:
                        raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} does not fit {tuple(dst.shape)}")
                    loaded += 1
        return {"loaded": loaded, "skipped": skipped}

    @property
    def transformer_layer_cls(self) -> Type[nn.Module]:
        return LlamaDecoderLayer

    @property
    def half_precision_dtype(self) -> torch.dtype:
        return torch.bfloat16

    @property
    def embed_dim(self) -> int:
        return self.llm.config.hidden_size

    @property
    def pad_token_id(self) -> int:
        return self.tokenizer.pad_token_id

    @property
    def prompt_builder_fn(self):
        """llama2.py:81-91: the pure builder for `llama2-*-pure` identifiers (the shipped `llama2-7b-pure`)."""
        if self.identifier.startswith("llama2-") and self.identifier.endswith("-pure"):
            from .data_utils import PurePromptBuilder
            return PurePromptBuilder
        raise ValueError(f"No PromptBuilder defined for LLM Backbone `{self.identifier}` (only the `-pure` Llama-2 builder is built)")

    def get_fsdp_wrapping_policy(self) -> Callable:
        cls = self.transformer_layer_cls
        return lambda module: isinstance(module, cls)

    def enable_gradient_checkpointing(self) -> None:
        """base_llm.py: gradient checkpointing on the HF model == full recompute inside each decoder layer here."""
        self.llm.config.activation_save_level = 0
        self.llm.config.activation_save_levels = None

    def set_activation_policy(self, keep_layers: int, keep_level: int = 1, rest_level: int = 0) -> None:
        """Mixed activation policy for 288 GB parts: the LAST `keep_layers` decoder layers keep their activations (`keep_level`: 1, 3
        or 2 -- they are the first to be consumed and freed by the backward), the others run at `rest_level` (0 = the reference's
        activation checkpointing, training/strategies/fsdp.py:211-223). Results are bit-identical for every mix."""
        n = self.llm.config.num_hidden_layers
        k = max(0, min(n, int(keep_layers)))
        self.llm.config.activation_save_levels = tuple([rest_level] * (n - k) + [keep_level] * k)

    def embed_input_ids(self, input_ids: torch.LongTensor) -> torch.Tensor:
        return self.llm.model.embed(input_ids)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, **contrastive_kwargs
                ) -> CausalLMOutputWithPast:
        """base_llm.py:198-241: passes the 9 extra contrastive kwargs through to LlamaForCausalLM.forward."""
        return self.llm(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                        past_key_values=past_key_values, inputs_embeds=inputs_embeds, labels=labels, use_cache=use_cache,
                        output_attentions=output_attentions, output_hidden_states=output_hidden_states, return_dict=return_dict,
                        **contrastive_kwargs)
