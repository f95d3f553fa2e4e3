"""Prefix-cached epsilon model for the diffusion samplers (round 6).

Reference: `MLA.predict_action_diff` (models/mla/model_mla.py:592-775) hands `self.vlm.forward` to `ddim_sample_loop`
(models/diffusion/gaussian_diffusion.py:608-688), which calls it once per DDIM step -- 8 whole 548-token forwards through the encoders and
the 32 decoder layers, although with causal attention everything in front of the `[t, x]` tokens
(`[BOS | pc 256 | img 256 | tac 1 | text[1:k] | proprio]`, models/vlm/prismatic.py:981-1038) is identical in all 8 and the token behind
them (`text[k:]`) cannot influence the rows that are read out (`noise_pred = final_layer(h_last)[k'+2 : k'+2+T]`, :1115-1126).

`PrefixCachedEps(vlm, **model_kwargs)` runs the encoders and ONE prefill over the prefix rows with the training kernels
(`ops.DecoderLayerFn._fwd`: fused RMSNorm / QKV + RoPE GEMM / flash attention / SwiGLU GEMMs), keeps every layer's packed post-RoPE
q|k|v rows, and then serves each `model(x, t)` call with a pass over the `1 + T` suffix rows per sample: skinny weight-streaming GEMMs
(`mla_gemv_bf16`, every weight read once per pass), `mla_attn_decode` against the cached keys / values, the same RMSNorm / RoPE / SwiGLU
kernels' arithmetic as training (RMSNorm and SwiGLU are applied inside the projections' input staging). The 6 x 32 launches of a pass
(5 with the rotary embedding in the QKV kernel's epilogue) are captured once into a HIP graph and replayed per DDIM step.

Semantics vs the reference: identical arithmetic up to summation order (fp32 accumulation everywhere), with ONE stated difference -- the
reference's point tokenizer draws fresh random FPS start indices inside every one of the 8 forwards (Point_PN.py:10); here they are
drawn once per action chunk (the prefix is computed once). With given start indices (`fps_starts_override`, as in
tests/test_inference_gpu.py) the two are the same function."""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from . import hip, ops

_USE_GRAPH = os.environ.get("MLA_INFER_GRAPH", "1") != "0"


class PrefixCachedEps:
    """One engine per (batch, prefix length, action rows): the per-layer q|k|v cache, the suffix pass's static input / output rows and its
    captured graph live as long as the engine, `prefill()` refreshes the cache for a new observation (the graph stays valid: same
    addresses). `PrefixCachedEps.for_inputs(vlm, ...)` returns the vlm's engine for the given inputs, prefilled."""

    @staticmethod
    def _splice_position(input_ids):
        tag_0 = 29871                                                        # prismatic.py:882-887 (eval)
        L = input_ids.shape[1]
        is_tag = input_ids == tag_0
        if not bool(is_tag.any(dim=1).all()):
            raise IndexError(f"input_ids row without the splice tag {tag_0} (models/vlm/prismatic.py:983)")
        k = (L - 1 - torch.flip(is_tag, dims=[1]).int().argmax(dim=1))       # last occurrence per row
        if not bool((k == k[0]).all()):
            raise ValueError("PrefixCachedEps needs the same splice position in every row (predict_action_diff is batch 1)")
        return int(k[0])

    @classmethod
    def for_inputs(cls, vlm, input_ids, n_action_rows: int = 1, **model_kwargs):
        k = cls._splice_position(input_ids)
        engines = vlm.__dict__.setdefault("_prefix_engines", {})
        key = (int(input_ids.shape[0]), k, int(n_action_rows), str(input_ids.device))
        eng = engines.get(key)
        if eng is None:
            if len(engines) >= 4:                                             # a handful of prompt lengths per process; each engine holds 0.4 GB at 7B
                engines.pop(next(iter(engines)))
            eng = engines[key] = cls(vlm, n_action_rows)
        eng.prefill(input_ids, k, **model_kwargs)
        return eng

    def __init__(self, vlm, n_action_rows: int = 1):
        self.vlm = vlm
        llm = vlm.llm_backbone.llm
        self.model, self.cfg = llm.model, llm.config
        self.T = n_action_rows
        self.R = 1 + self.T
        self.nheads, self.eps = self.cfg.num_attention_heads, self.cfg.rms_norm_eps
        self.cache = None
        self.graph = None
        self._graph_failed = False
        self._packed = None          # per layer: the 9 weights with q|k|v and gate|up as views of ONE buffer each (see _weights)
        self._packed_key = None

    def _weights(self):
        """Every layer's (ln1, wq, wk, wv, wo, ln2, wg, wu, wd) with q|k|v and gate|up adjacent in memory, so that the prefill runs the
        fused QKV + RoPE and gate|up + SwiGLU GEMMs and a suffix pass needs one GEMV each instead of three / two (33 MB projections are
        ~40 % launch + ramp). Under FSDPStrategy the parameters already live like that in the unit's flat buffer (views are used as they
        are); otherwise the engine keeps packed COPIES (9.4 GB at 7B), rebuilt when a parameter's storage or version changes."""
        key = tuple((p.data_ptr(), p._version) for layer in self.model.layers for p in layer._weights())
        shared = self.vlm.__dict__.setdefault("_prefix_packed", {})          # one packed copy per model, shared by its engines
        if shared.get("key") != key:
            packed = []
            with torch.no_grad(), torch.inference_mode(False):
                for layer in self.model.layers:
                    ln1, wq, wk, wv, wo, ln2, wg, wu, wd = layer._weights()
                    if ops.cat_view((wq, wk, wv)) is None:
                        buf = torch.cat([wq.detach(), wk.detach(), wv.detach()], 0)
                        H = wq.shape[0]
                        wq, wk, wv = buf[:H], buf[H:H + wk.shape[0]], buf[H + wk.shape[0]:]
                    if ops.cat_view((wg, wu)) is None:
                        buf = torch.cat([wg.detach(), wu.detach()], 0)
                        wg, wu = buf[:wg.shape[0]], buf[wg.shape[0]:]
                    packed.append((ln1, wq, wk, wv, wo, ln2, wg, wu, wd))
            shared["key"], shared["packed"] = key, packed
        if self._packed is not shared["packed"]:
            self.graph = None            # a captured pass holds the previous buffers' addresses
            self._packed, self._packed_key = shared["packed"], key
        return self._packed

    def prefill(self, input_ids, k, images=None, point_cloud=None, camera_name=None, proprio=None, tactile=None, gripper_xyz=None, **unused):
        vlm, bf16, dev = self.vlm, torch.bfloat16, input_ids.device
        with torch.no_grad():
            parts, _, _, _, _, _ = vlm.get_fused_tokens(images, point_cloud, tactile, gripper_xyz, camera_name)
            vlm.vision_tower_2d.assert_masks_ok()
            text_emb = vlm.llm_backbone.embed_input_ids(input_ids)
            proprio_e = vlm.proprio_embedder(proprio.to(bf16))
            prefix = torch.cat([text_emb[:, :1]] + parts + [text_emb[:, 1:k], proprio_e], dim=1).contiguous()       # [B, S_p, H]
            B, S_p, H = prefix.shape
            if self.cache is None:
                self.B, self.S_p, self.H = B, S_p, H
                self.S_cap = S_p + self.R
                self.D = H // self.nheads
                rot = self.model.layers[0].self_attn.rotary_emb
                self.cos_p, self.sin_p = rot.tables(S_p, dev)
                cos_c, sin_c = rot.tables(self.S_cap, dev)
                self.cos_s, self.sin_s = cos_c[S_p:].contiguous(), sin_c[S_p:].contiguous()
                with torch.inference_mode(False):                            # the engine outlives the (inference-mode) call that creates it
                    self.cache = [torch.empty((B, self.S_cap, 3 * H), dtype=bf16, device=dev) for _ in self.model.layers]
                    self.h_in = torch.zeros((B * self.R, H), dtype=bf16, device=dev)
                    self.h_out = torch.zeros((B * self.R, H), dtype=bf16, device=dev)
            assert (B, S_p, H) == (self.B, self.S_p, self.H)
            # ---- prefill: the training forward kernels, one layer at a time; keep the packed post-RoPE q|k|v rows
            h = prefix.reshape(B * S_p, H)
            for w, c in zip(self._weights(), self.cache):
                h, saved = ops.DecoderLayerFn._fwd(h, None, self.cos_p, self.sin_p, B, S_p, self.nheads, self.eps, w)
                c[:, :S_p].copy_(saved[2][:B * S_p].view(B, S_p, 3 * H))
                del saved

    # ------------------------------------------------------------------------------------------ one pass over the suffix rows
    def _gemv(self, x, weights, out=None, residual=None, rpb=1, out_bs=0, **pre):
        """f(x) [M, K] @ cat(weights)^T (+ residual) -> [M, sum N]; one launch when the weights are adjacent in memory.
        pre: norm_weight= / eps= (RMSNorm of the rows) or swiglu=True (x = packed gate|up rows), applied inside the kernel's input staging."""
        M = x.shape[0]
        wcat = ops.cat_view(weights) if len(weights) > 1 else weights[0]
        N = sum(w.shape[0] for w in weights)
        if out is None:
            out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
            rpb, out_bs = M, 0
        ldo = out.stride(-2)
        if wcat is not None:
            hip.gemv(x, wcat, out, ldo, out_bs, rpb, residual, **pre)
        else:                                                                # parameters not laid out back to back (no FlatUnit): one launch each
            off = 0
            for w in weights:
                hip.gemv(x, w, out, ldo, out_bs, rpb, None if residual is None else residual[:, off:], out_col=off, **pre)
                off += w.shape[0]
        return out

    def _suffix_pass(self):
        B, R, H, S_p, S_cap = self.B, self.R, self.H, self.S_p, self.S_cap
        h = self.h_in
        scale = 1.0 / math.sqrt(self.D)
        for (ln1, wq, wk, wv, wo, ln2, wg, wu, wd), c in zip(self._packed, self.cache):
            # north_star's "fused RMSNorm + RoPE + QKV" as ONE kernel: RMSNorm inside the projection's input staging, the rotary embedding of
            # the q and k columns in its epilogue; the rows go straight into the cache slots [S_p, S_p + R) of every sample
            fused = ops.cat_view((wq, wk, wv)) is not None and self.D == 128
            self._gemv(h, (wq, wk, wv), out=c[:, S_p:], rpb=R, out_bs=c.stride(0), norm_weight=ln1, eps=self.eps,
                       **({"rope": (self.cos_s, self.sin_s, 2 * H)} if fused else {}))
            if not fused:
                for b in range(B):
                    hip.rope_inplace(c[b, S_p:], self.cos_s, self.sin_s, R, self.nheads, self.D, 0, H)
            o = hip.attn_decode(c, B, self.nheads, self.D, S_cap, R, scale)
            h1 = self._gemv(o, (wo,), residual=h)
            gu = self._gemv(h1, (wg, wu), norm_weight=ln2, eps=self.eps)
            h = self._gemv(gu, (wd,), residual=h1, swiglu=True)                # SwiGLU inside the down projection's input staging
        hn, _ = hip.rmsnorm_fwd(h, self.model.norm.weight, self.eps)
        self.h_out.copy_(hn)

    def _run(self):
        if _USE_GRAPH and not self._graph_failed:
            if self.graph is None:
                try:
                    self._suffix_pass()                                      # warm-up outside the capture (function attributes, allocator)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._suffix_pass()
                    self.graph = g
                except Exception:   # noqa: BLE001 -- a failed capture is not fatal: the eager launches compute the same thing
                    self._graph_failed = True
                    self.graph = None
                    torch.cuda.synchronize()
            if self.graph is not None:
                self.graph.replay()
                return
        self._suffix_pass()

    # ------------------------------------------------------------------------------------------ the model(x, t, **kw) the samplers call
    def __call__(self, x, t, **ignored):
        """Same contract as PrismaticVLM.forward in eval mode: returns (None, noise_pred [B, T, action_dim])."""
        vlm, bf16 = self.vlm, torch.bfloat16
        with torch.no_grad():
            x_e = vlm.x_embedder(x.to(bf16))                                  # [B, T, H]   (prismatic.py:873-880 casts)
            t_e = vlm.t_embedder(t.to(bf16)).unsqueeze(1)                     # [B, 1, H]
            assert x_e.shape[1] == self.T, (x_e.shape, self.T)
            self.h_in.copy_(torch.cat([t_e, x_e], dim=1).reshape(self.B * self.R, self.H))
            self._run()
            picked = self.h_out.view(self.B, self.R, self.H)[:, 1:].reshape(self.B * self.T, self.H).contiguous()
            noise_pred = vlm.final_layer(picked).view(self.B, self.T, -1)
        return None, noise_pred
