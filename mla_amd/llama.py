"""Drop-in Llama-2 decoder for the MLA hot path, executing on the hand-written gfx950 kernels.

Mirrors the API surface of the reference's vendored transformers/models/llama/modeling_llama.py (class names,
constructor/forward signatures, attribute and state-dict names) for the pieces scripts/train.py and training/strategies
touch: LlamaRMSNorm (:76-90), LlamaRotaryEmbedding (:96-145), LlamaMLP (:211-242), LlamaAttention / LlamaFlashAttention2
(:257-597), LlamaDecoderLayer (:695-767, the FSDP / activation-checkpoint unit), LlamaModel (:912-1127) and the modified
LlamaForCausalLM (:1129-1317, contrastive tap on hidden_states[8]).

Scope: training/prefill forward + backward in bf16 (no KV cache / generation; pretraining_tp == 1; MHA with
head_dim 128, which is what Llama-2-7B uses). Parameters are created fp32 like the reference's load path and are switched
to bf16 compute storage by mla_amd.fsdp (or ``model.to(torch.bfloat16)`` for ad-hoc use).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import hip, ops
from .fuser import CoordinateAwareContrastiveLoss, TactileContrastiveLoss
from .modeling_outputs import CausalLMOutputWithPast


@dataclass
class LlamaConfig:
    """Subset of transformers LlamaConfig (configuration_llama.py) used on this path; defaults = Llama-2-7b."""
    vocab_size: int = 32000
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: Optional[int] = None
    hidden_act: str = "silu"
    max_position_embeddings: int = 4096
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    pad_token_id: Optional[int] = None
    bos_token_id: int = 1
    eos_token_id: int = 2
    pretraining_tp: int = 1
    attention_dropout: float = 0.0
    attention_bias: bool = False
    use_cache: bool = False
    output_hidden_states: bool = False
    use_return_dict: bool = True
    # mla_amd extensions
    activation_save_level: int = 2     # 2 keep all, 1 recompute cheap elementwise, 3 = 1 without the kept SwiGLU product, 0 full recompute
    activation_save_levels: Optional[tuple] = None   # per-layer override (mixed policy: keep as many layers as fit in HBM, checkpoint the rest)
    contrastive_tap_layer: int = 8     # index into hidden_states (reference hard-codes 8, modeling_llama.py:1274)
    compute_lm_logits: bool = True     # reference always materialises fp32 logits + CE even when unused (:1255-1269)
    lazy_lm_head: bool = True          # round 6: in training they are computed on first access of output.logits / output.loss
                                       # (mla_amd/modeling_outputs.py); False = inside forward, like the reference

    def layer_save_level(self, layer_idx: int) -> int:
        lv = self.activation_save_levels
        return self.activation_save_level if lv is None else int(lv[layer_idx])

    def __post_init__(self):
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        if self.num_key_value_heads != self.num_attention_heads:
            raise NotImplementedError("GQA is not on the MLA-Llama2-7B path (num_key_value_heads must equal heads)")
        if self.hidden_size // self.num_attention_heads != 128:
            raise NotImplementedError("the attention kernels are built for head_dim 128 (Llama-2-7B)")
        if self.pretraining_tp != 1:
            raise NotImplementedError("pretraining_tp > 1 weight slicing is not supported")


class Linear(nn.Linear):
    """nn.Linear whose forward/backward are the MFMA GEMMs (same parameters / state-dict keys)."""

    def forward(self, x, residual=None):  # noqa: D102
        return ops.linear(x, (self.weight,), self.bias, residual)


class LlamaRMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states):
        return ops.rmsnorm(hidden_states, self.weight, self.variance_epsilon)


class LlamaRotaryEmbedding(nn.Module):
    """cos/sin tables in fp32 (modeling_llama.py:96-145); computed once per (seq_len, device) on the host."""

    def __init__(self, dim, max_position_embeddings=2048, base=10000.0, device=None, scaling_factor=1.0):
        super().__init__()
        self.dim, self.base = dim, base
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
        self.register_buffer("inv_freq", inv_freq, persistent=False)
        self._cache = {}

    def tables_for_positions(self, positions: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """cos / sin rows for explicit position ids [S] (shared-prefix sequences: the suffix groups repeat the same positions);
        the arithmetic of `tables` (fp32 outer product, modeling_llama.py:131-145)."""
        inv_freq = 1.0 / (self.base ** (torch.arange(0, self.dim, 2, dtype=torch.int64).float() / self.dim))
        freqs = torch.outer(positions.detach().cpu().to(torch.float32), inv_freq)
        return freqs.cos().contiguous().to(positions.device), freqs.sin().contiguous().to(positions.device)

    def tables(self, seq_len: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
        key = (seq_len, str(device))
        if key not in self._cache:
            inv_freq = 1.0 / (self.base ** (torch.arange(0, self.dim, 2, dtype=torch.int64).float() / self.dim))
            freqs = torch.outer(torch.arange(seq_len, dtype=torch.float32), inv_freq)
            self._cache[key] = (freqs.cos().contiguous().to(device), freqs.sin().contiguous().to(device))
        return self._cache[key]


class LlamaMLP(nn.Module):
    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.config = config
        self.hidden_size, self.intermediate_size = config.hidden_size, config.intermediate_size
        self.gate_proj = Linear(self.hidden_size, self.intermediate_size, bias=False)
        self.up_proj = Linear(self.hidden_size, self.intermediate_size, bias=False)
        self.down_proj = Linear(self.intermediate_size, self.hidden_size, bias=False)


class LlamaAttention(nn.Module):
    """Parameter container with the reference's names; the math runs inside ops.DecoderLayerFn."""

    def __init__(self, config: LlamaConfig, layer_idx: Optional[int] = None):
        super().__init__()
        self.config, self.layer_idx = config, layer_idx
        self.hidden_size, self.num_heads = config.hidden_size, config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.q_proj = Linear(self.hidden_size, self.num_heads * self.head_dim, bias=False)
        self.k_proj = Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.v_proj = Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=False)
        self.o_proj = Linear(self.hidden_size, self.hidden_size, bias=False)
        self.rotary_emb = LlamaRotaryEmbedding(self.head_dim, config.max_position_embeddings, config.rope_theta)


LlamaFlashAttention2 = LlamaAttention  # the constructor flag use_flash_attention_2 selects the same HIP kernel
LlamaSdpaAttention = LlamaAttention


class LlamaDecoderLayer(nn.Module):
    """FSDP / checkpoint unit (models/backbones/llm/llama2.py:93-95). forward(hidden_states, attention_mask=...)."""

    def __init__(self, config: LlamaConfig, layer_idx: int):
        super().__init__()
        self.config = config
        self.hidden_size = config.hidden_size
        self.layer_idx = layer_idx
        self.self_attn = LlamaAttention(config, layer_idx)
        self.mlp = LlamaMLP(config)
        self.input_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self._grad_hook = None  # set by mla_amd.fsdp: fires when this layer's backward has been enqueued
        self._gather_wait = None  # set by mla_amd.fsdp (world > 1): makes the current stream wait for this layer's all-gather
        self._fold_out = None   # ops.NormFoldIO.out of the last forward (x * g and sum(x^2) partials for the NEXT layer's input norm)

    def _weights(self):
        a, m = self.self_attn, self.mlp
        return (self.input_layernorm.weight, a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.o_proj.weight,
                self.post_attention_layernorm.weight, m.gate_proj.weight, m.up_proj.weight, m.down_proj.weight)

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, output_attentions=False,
                use_cache=False, cache_position=None, seqlens: Optional[torch.Tensor] = None, rope_tables=None, norm_fold=None, **kwargs):
        """norm_fold = (what the previous layer's down projection made for this layer's input norm or None, the next LlamaDecoderLayer
        or None): the RMSNorms are folded into the projections (ops.NormFoldIO); None = the separate rmsnorm launches."""
        if past_key_value is not None or use_cache or output_attentions:
            raise NotImplementedError("KV cache / attention-weight output are inference features (SURVEY 8f rank 2)")
        B, S, _ = hidden_states.shape
        if seqlens is None and attention_mask is not None:
            seqlens = attention_mask.reshape(B, -1).sum(-1).to(torch.int32)
        cos, sin = rope_tables if rope_tables is not None else self.self_attn.rotary_emb.tables(S, hidden_states.device)
        h = ops.unit_boundary(hidden_states, self._grad_hook)
        io = None
        if norm_fold is not None:
            carry, nxt = norm_fold
            next_ln = None
            if nxt is not None:
                if nxt._gather_wait is not None:
                    nxt._gather_wait()           # this layer's down projection reads the NEXT layer's input_layernorm weight
                next_ln = nxt.input_layernorm.weight
            io = ops.NormFoldIO(pre=carry, next_ln=next_ln)
        out = ops.decoder_layer(h, seqlens, cos, sin, self.config.num_attention_heads, self.config.rms_norm_eps,
                                self.config.layer_save_level(self.layer_idx), self._weights(), fold_io=io)
        self._fold_out = io.out if io is not None else None
        return (out,)

    def forward_readout(self, hidden_states, rows, seqlens: Optional[torch.Tensor] = None, rope_tables=None):
        """Rows `rows` (flat indices into B * S) of this layer's output [n, H]: the row-wise half of the layer (o_proj, MLP) runs on
        those rows only (ops.ReadoutLayerFn). For the LAST decoder layer when nothing else of its output is read."""
        B, S, _ = hidden_states.shape
        cos, sin = rope_tables if rope_tables is not None else self.self_attn.rotary_emb.tables(S, hidden_states.device)
        h = ops.unit_boundary(hidden_states, self._grad_hook)
        return ops.decoder_layer_readout(h, seqlens, cos, sin, self.config.num_attention_heads, self.config.rms_norm_eps, rows,
                                         self._weights())


class LlamaModel(nn.Module):
    def __init__(self, config: LlamaConfig):
        super().__init__()
        self.config = config
        self.padding_idx, self.vocab_size = config.pad_token_id, config.vocab_size
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, self.padding_idx)
        self.layers = nn.ModuleList([LlamaDecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = LlamaRMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.gradient_checkpointing = False

    def get_input_embeddings(self):
        return self.embed_tokens

    def embed(self, input_ids):
        return ops.embedding(input_ids, self.embed_tokens.weight)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None, cache_position=None,
                attn_groups=None, readout_rows=None):
        """readout_rows (int64 [n], flat indices into B * S; round 6, opt-in): only these rows of the final hidden state are wanted --
        the last decoder layer runs its row-wise half on them alone (ops.ReadoutLayerFn) and the call returns
        (norm(rows) [n, H], hidden states of the layers in front, a callable that produces the dense final hidden state on demand).
        attn_groups = (first suffix row, rows per group) with position_ids [S] (round 6, opt-in): one sequence per sample laid out
        as [prefix | R suffix groups]; every suffix row attends to the prefix and, causally, to its own group, and carries the position
        it has in the reference's R separate sequences (models/mla/model_mla.py:148-180 tiles the whole sample R times)."""
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You cannot specify both input_ids and inputs_embeds at the same time, and must specify either one")
        if past_key_values is not None or use_cache:
            raise NotImplementedError("KV cache is an inference feature (SURVEY 8f rank 2)")
        if inputs_embeds is None:
            inputs_embeds = self.embed(input_ids)
        B, S, _ = inputs_embeds.shape
        # right-padded batches: per-row valid length (flash varlen semantics; positions stay arange(S), :985-990)
        seqlens = None if attention_mask is None else attention_mask.reshape(B, S).sum(-1).to(torch.int32)
        hidden_states = inputs_embeds
        all_hidden = () if output_hidden_states else None
        rope_tables = None
        if attn_groups is not None:
            if position_ids is None:
                raise ValueError("attn_groups needs explicit position_ids ([S], or [B, S] with per-sample group starts)")
            # position_ids [S]: one table for every sample; [B, S]: per-sample positions (ragged prompts) -> tables [B * S, 64]
            rope_tables = self.layers[0].self_attn.rotary_emb.tables_for_positions(position_ids.reshape(-1))
            assert rope_tables[0].shape[0] in (S, B * S)
        with ops.attn_groups(attn_groups):
            fold, carry, n_layers = ops.norm_fold_enabled(), None, len(self.layers)
            for i, layer in enumerate(self.layers):
                if output_hidden_states:
                    all_hidden += (hidden_states,)
                if readout_rows is not None and i == n_layers - 1:
                    last_in = hidden_states
                    picked = self.norm(layer.forward_readout(last_in, readout_rows, seqlens=seqlens, rope_tables=rope_tables))

                    def dense_last(layer=layer, last_in=last_in):
                        with ops.attn_groups(attn_groups):
                            return self.norm(layer(last_in, seqlens=seqlens, rope_tables=rope_tables)[0])
                    return picked, all_hidden, dense_last
                if fold:       # RMSNorms folded into the projections: layer i's down projection prepares layer i + 1's input norm
                    nxt = self.layers[i + 1] if (i + 1 < n_layers and not (readout_rows is not None and i + 1 == n_layers - 1)) else None
                    hidden_states = layer(hidden_states, seqlens=seqlens, rope_tables=rope_tables, norm_fold=(carry, nxt))[0]
                    carry, layer._fold_out = layer._fold_out, None
                else:
                    hidden_states = layer(hidden_states, seqlens=seqlens, rope_tables=rope_tables)[0]
        hidden_states = self.norm(hidden_states)
        if output_hidden_states:
            all_hidden += (hidden_states,)
        return hidden_states, all_hidden


class LlamaForCausalLM(nn.Module):
    """modeling_llama.py:1129-1317 incl. the MLA additions (contrastive heads built by default, :1133-1156)."""
    _tied_weights_keys = ["lm_head.weight"]

    def __init__(self, config: LlamaConfig, use_token_contrastive_loss: bool = True, use_tactile_contrastive_loss: bool = True,
                 contrastive_projection_dim: int = 256):
        super().__init__()
        self.config = config
        self.model = LlamaModel(config)
        self.vocab_size = config.vocab_size
        self.lm_head = Linear(config.hidden_size, config.vocab_size, bias=False)
        self.use_token_contrastive_loss = use_token_contrastive_loss
        if use_token_contrastive_loss:
            self.coordinate_aware_contrastive_loss_module = CoordinateAwareContrastiveLoss(
                feature_dim=config.hidden_size, projection_dim=contrastive_projection_dim)
        self.use_tactile_contrastive_loss = use_tactile_contrastive_loss
        if use_tactile_contrastive_loss:
            self.tactile_contrastive_loss_module = TactileContrastiveLoss(
                feature_dim=config.hidden_size, projection_dim=contrastive_projection_dim)
        self.post_init()

    def post_init(self):
        """HF _init_weights (normal(0, initializer_range=0.02) on Linear/Embedding)."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, mean=0.0, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, mean=0.0, std=0.02)

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def set_input_embeddings(self, value):
        self.model.embed_tokens = value

    def get_output_embeddings(self):
        return self.lm_head

    def set_output_embeddings(self, new_embeddings):
        self.lm_head = new_embeddings

    def get_decoder(self):
        return self.model

    def resize_token_embeddings(self, new_num_tokens: int, pad_to_multiple_of: Optional[int] = None):
        """PreTrainedModel.resize_token_embeddings (transformers/modeling_utils.py:1876) for embed_tokens + lm_head:
        keeps the old rows, new rows ~ N(0, 0.02) (scripts/train.py:143-155 then overwrites them with the mean)."""
        if pad_to_multiple_of:
            new_num_tokens = ((new_num_tokens + pad_to_multiple_of - 1) // pad_to_multiple_of) * pad_to_multiple_of
        old = self.model.embed_tokens
        if new_num_tokens == old.num_embeddings:
            return old
        H = self.config.hidden_size
        emb = nn.Embedding(new_num_tokens, H, self.config.pad_token_id).to(old.weight.device, old.weight.dtype)
        nn.init.normal_(emb.weight, std=0.02)
        head = Linear(H, new_num_tokens, bias=False).to(old.weight.device, old.weight.dtype)
        nn.init.normal_(head.weight, std=0.02)
        n = min(new_num_tokens, old.num_embeddings)
        with torch.no_grad():
            emb.weight[:n] = old.weight[:n]
            head.weight[:n] = self.lm_head.weight[:n]
        self.model.embed_tokens, self.lm_head = emb, head
        self.config.vocab_size = self.vocab_size = self.model.vocab_size = new_num_tokens
        return emb

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, output_attentions=None, output_hidden_states=None, return_dict=None,
                cache_position=None, pc_token_indices=None, img_token_indices=None, tac_token_indices=None,
                patch_correspondence_indices=None, correspondence_valid_mask=None, positive_pc_indices_for_tac=None,
                linear_positive_img_indices_for_tac=None, compute_token_contrastive_loss: bool = False,
                compute_tactile_contrastive_loss: bool = False, attn_groups=None, readout_rows=None):
        """readout_rows (round 6, opt-in): the caller reads only these rows of the final hidden state (the action read-out of the
        diffusion branch). Honoured in training with the lazy lm_head: `output.readout_hidden` [n, H] holds them, the dense final hidden
        state / logits / loss are produced on first access of `output.hidden_states` / `.logits` / `.loss`. Otherwise the forward is
        the dense one and `readout_hidden` is a gather of its rows."""
        output_hidden_states = (output_hidden_states if output_hidden_states is not None else self.config.output_hidden_states)
        need_tap = self.training and (compute_token_contrastive_loss or compute_tactile_contrastive_loss)
        use_readout = readout_rows is not None and self.training and self.config.lazy_lm_head and attn_groups is None
        model_out = self.model(input_ids=input_ids, attention_mask=attention_mask, inputs_embeds=inputs_embeds,
                               past_key_values=past_key_values, use_cache=use_cache,
                               output_hidden_states=bool(output_hidden_states or need_tap),
                               position_ids=position_ids if attn_groups is not None else None, attn_groups=attn_groups,
                               readout_rows=readout_rows if use_readout else None)
        readout_hidden, lazy_last = None, None
        if use_readout:
            readout_hidden, all_hidden, dense_last = model_out
            cell = {}

            def lazy_last():
                if "v" not in cell:
                    cell["v"] = dense_last()
                return cell["v"]
            ref = inputs_embeds if inputs_embeds is not None else input_ids
            B, S, H = ref.shape[0], ref.shape[1], self.config.hidden_size
            hidden_states = None
        else:
            hidden_states, all_hidden = model_out
            B, S, H = hidden_states.shape
            if readout_rows is not None:
                readout_hidden = ops.gather_rows(hidden_states.reshape(B * S, H), readout_rows)
        logits, loss, lazy_lm = None, None, None
        if self.config.compute_lm_logits or labels is not None:
            def lm_and_ce(hidden_states=hidden_states, labels=labels):
                if hidden_states is None:
                    hidden_states = lazy_last()              # read-out forward: the dense last layer runs now, with autograd
                h2 = hidden_states.reshape(B * S, H)
                lg = LMHeadFn.apply(h2, self.lm_head.weight).view(B, S, -1)  # fp32, = lm_head(h).float()  (:1254-1255)
                ce = None
                if labels is not None:
                    # shift so that tokens < n predict n (:1258-1262): instead of slicing the 2.25 GB logits tensor the
                    # labels are shifted left and the last position ignored -- the same set of (row, label) pairs
                    shifted = torch.full_like(labels, -100)
                    shifted[:, :-1] = labels[:, 1:]
                    ce = ops.cross_entropy(lg.view(B * S, -1), shifted.reshape(-1))
                return lg, ce
            if self.config.lazy_lm_head and self.training:
                # Round 6: the diffusion objective never reads logits / CE and the trainer drops `output`
                # (base_strategy_mla.py:307,334) -- they are produced on first access of output.logits / output.loss instead
                lazy_lm = lm_and_ce
            else:
                logits, loss = lm_and_ce()

        img_pc_contrastive_loss = None
        if self.training and compute_token_contrastive_loss:
            tap = all_hidden[self.config.contrastive_tap_layer]
            pc_start, pc_end = pc_token_indices
            img_start, img_end = img_token_indices
            img_pc_contrastive_loss = self.coordinate_aware_contrastive_loss_module(
                image_features=tap[:, img_start:img_end, :], pointcloud_features=tap[:, pc_start:pc_end, :],
                patch_indices=patch_correspondence_indices, valid_mask=correspondence_valid_mask)
            if lazy_lm is None:
                loss = loss + img_pc_contrastive_loss
        tactile_contrastive_loss = None
        if self.training and compute_tactile_contrastive_loss:
            tap = all_hidden[self.config.contrastive_tap_layer]
            pc_start, pc_end = pc_token_indices
            img_start, img_end = img_token_indices
            tac_start, tac_end = tac_token_indices
            tactile_contrastive_loss = self.tactile_contrastive_loss_module(
                tac_features=tap[:, tac_start:tac_end, :], pc_features=tap[:, pc_start:pc_end, :],
                img_features=tap[:, img_start:img_end, :], positive_pc_indices=positive_pc_indices_for_tac,
                linear_positive_img_indices=linear_positive_img_indices_for_tac)
            if lazy_lm is None:
                loss = loss + tactile_contrastive_loss
        return CausalLMOutputWithPast(loss=loss, logits=logits, img_pc_contrastive_loss=img_pc_contrastive_loss,
                                      tactile_contrastive_loss=tactile_contrastive_loss, past_key_values=None,
                                      hidden_states=all_hidden if output_hidden_states else None, attentions=None, lazy_lm=lazy_lm,
                                      readout_hidden=readout_hidden, lazy_last=lazy_last if (use_readout and output_hidden_states) else None)


class LMHeadFn(torch.autograd.Function):
    """logits = (h @ W^T) written as fp32 straight from the GEMM epilogue (no bf16 logits + .float() copy)."""

    @staticmethod
    def forward(ctx, h2, weight):
        ctx.save_for_backward(h2)
        ctx.weight = weight
        return hip.gemm(h2, weight, out_dtype=torch.float32)

    @staticmethod
    def backward(ctx, dlogits):
        (h2,) = ctx.saved_tensors
        w = ctx.weight
        d = dlogits if dlogits.dtype == torch.bfloat16 else hip.cast_f32_to_bf16(dlogits.contiguous())
        d = d.contiguous()
        dh = hip.gemm(d, w, b_mode=1) if ctx.needs_input_grad[0] else None
        dw = ops.deliver_wgrad((w,), d, h2, (ctx.needs_input_grad[1],))[0]
        return dh, dw
