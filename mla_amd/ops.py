"""torch.autograd.Function wrappers around the C-ABI kernels (mla_amd/hip.py).

Conventions
* activations and compute weights are bf16 CUDA tensors; reductions / statistics / master gradients are fp32;
* weight gradients go straight from the wgrad GEMM epilogue into ``param.main_grad`` (an fp32 view into the unit's flat
  gradient buffer, installed by mla_amd.fsdp) -- first write of a step overwrites (no zero-fill pass), later writes
  accumulate. Parameters without ``main_grad`` (unit tests, ad-hoc use) get a regular ``.grad`` tensor instead;
* there is no CPU path: every op raises if handed a CPU tensor or if libmla_hip.so is missing.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch

from . import hip

BF16 = torch.bfloat16


# ------------------------------------------------------------------------------------------------- helpers
def cat_view(tensors: Sequence[torch.Tensor]) -> Optional[torch.Tensor]:
    """If the 2-D tensors are contiguous, share K and lie back-to-back in one storage (flat-parameter layout),
    return the [sum N_i, K] view over all of them; else None."""
    if len(tensors) == 1:
        return tensors[0] if tensors[0].is_contiguous() else None
    t0 = tensors[0]
    K = t0.shape[1]
    ptr = t0.data_ptr()
    total = 0
    for t in tensors:
        if t.dim() != 2 or t.shape[1] != K or not t.is_contiguous() or t.data_ptr() != ptr or t.dtype != t0.dtype:
            return None
        if t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr():
            return None
        ptr += t.numel() * t.element_size()
        total += t.shape[0]
    return torch.as_strided(t0, (total, K), (K, 1))


def _touch(param) -> bool:
    """Returns whether main_grad already holds a contribution from this step (then accumulate) and marks it."""
    acc = bool(getattr(param, "_mg_touched", False))
    param._mg_touched = True
    return acc


def deliver_wgrad(weights: Sequence[torch.Tensor], dy2: torch.Tensor, x2: torch.Tensor, needs: Sequence[bool]):
    """dW_i = dy[:, slice_i]^T @ x for every weight; into main_grad when present, else returned as tensors."""
    grads: List[Optional[torch.Tensor]] = [None] * len(weights)
    mgs = [getattr(w, "main_grad", None) for w in weights]
    if all(m is not None for m in mgs) and all(needs):
        mcat = cat_view(mgs)
        states = {bool(getattr(w, "_mg_touched", False)) for w in weights}
        if mcat is not None and len(states) == 1:
            acc = states.pop()
            hip.gemm(dy2, x2, out=mcat, a_mode=1, b_mode=1, accumulate=acc)
            for w in weights:
                w._mg_touched = True
            return grads
    off = 0
    for i, w in enumerate(weights):
        n = w.shape[0]
        if needs[i]:
            dys = dy2[:, off:off + n]
            if mgs[i] is not None:
                hip.gemm(dys, x2, out=mgs[i], a_mode=1, b_mode=1, M=n, accumulate=_touch(w))
            else:
                grads[i] = hip.gemm(dys, x2, a_mode=1, b_mode=1, M=n, out_dtype=torch.float32).to(w.dtype)
        off += n
    return grads


def deliver_wgrad_nt(weights: Sequence[torch.Tensor], dyT: torch.Tensor, xT: torch.Tensor, needs: Sequence[bool]):
    """Same as deliver_wgrad with pre-transposed operands: dW_i = dyT[rows_i] @ xT^T  (dyT [sum N_i, T], xT [K, T], both
    k-contiguous -> the fast NT kernel)."""
    grads: List[Optional[torch.Tensor]] = [None] * len(weights)
    mgs = [getattr(w, "main_grad", None) for w in weights]
    if all(m is not None for m in mgs) and all(needs):
        mcat = cat_view(mgs)
        states = {bool(getattr(w, "_mg_touched", False)) for w in weights}
        if mcat is not None and len(states) == 1:
            hip.gemm(dyT, xT, out=mcat, accumulate=states.pop())
            for w in weights:
                w._mg_touched = True
            return grads
    off = 0
    for i, w in enumerate(weights):
        n = w.shape[0]
        if needs[i]:
            if mgs[i] is not None:
                hip.gemm(dyT[off:off + n], xT, out=mgs[i], accumulate=_touch(w))
            else:
                grads[i] = hip.gemm(dyT[off:off + n], xT, out_dtype=torch.float32).to(w.dtype)
        off += n
    return grads


def deliver_vec_grad(param: torch.Tensor, compute):
    """compute(out_f32, accumulate) fills a 1-D fp32 gradient. Routes to main_grad or returns a tensor."""
    mg = getattr(param, "main_grad", None)
    if mg is not None:
        compute(mg, _touch(param))
        return None
    g = torch.empty(param.shape, dtype=torch.float32, device=param.device)
    compute(g, False)
    return g.to(param.dtype)


def _as2d(x: torch.Tensor) -> torch.Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    return x2 if x2.is_contiguous() else x2.contiguous()


def _check_bf16_cuda(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("mla_amd ops run on the GPU only (no CPU fallback)")
        if t.dtype != BF16:
            raise TypeError(f"mla_amd ops expect bfloat16 tensors, got {t.dtype}")


# ------------------------------------------------------------------------------------------------- linear
class LinearFn(torch.autograd.Function):
    """y = x @ cat(W_i)^T (+ bias) (+ residual). One fused GEMM when the weights are adjacent in the flat buffer."""

    @staticmethod
    def forward(ctx, x, residual, bias, *weights):
        _check_bf16_cuda(x, residual, bias, *weights)
        x2 = _as2d(x)
        K = x2.shape[1]
        ntot = sum(w.shape[0] for w in weights)
        out = torch.empty((x2.shape[0], ntot), dtype=BF16, device=x.device)
        res2 = _as2d(residual) if residual is not None else None
        wcat = cat_view(weights)
        if wcat is not None:
            hip.gemm(x2, wcat, out=out, bias=bias, residual=res2)
        else:
            if bias is not None and len(weights) != 1:
                raise ValueError("bias is only supported with a single weight")
            off = 0
            for w in weights:
                n = w.shape[0]
                hip.gemm(x2, w.contiguous(), out=out[:, off:off + n], bias=bias,
                         residual=res2[:, off:off + n] if res2 is not None else None)
                off += n
        ctx.save_for_backward(x2)
        ctx.weights, ctx.bias, ctx.has_res = weights, bias, residual is not None
        ctx.in_shape = x.shape
        return out.view(*x.shape[:-1], ntot)

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        weights = ctx.weights
        dy2 = _as2d(dy)
        ni = ctx.needs_input_grad
        dx = None
        if ni[0]:
            wcat = cat_view(weights)
            if wcat is not None:
                dx = hip.gemm(dy2, wcat, b_mode=1)
            else:
                off = 0
                for w in weights:
                    n = w.shape[0]
                    part = hip.gemm(dy2[:, off:off + n], w.contiguous(), b_mode=1, K=n)
                    dx = part if dx is None else hip.add_bf16(dx, part)
                    off += n
            dx = dx.view(ctx.in_shape)
        wg = deliver_wgrad(weights, dy2, x2, ni[3:])
        db = None
        if ctx.bias is not None and ni[2]:
            db = deliver_vec_grad(ctx.bias, lambda out, acc: hip.colsum(dy2, out, acc))
        dres = dy if (ctx.has_res and ni[1]) else None
        return (dx, dres, db, *wg)


def linear(x, weights, bias=None, residual=None):
    if isinstance(weights, torch.Tensor):
        weights = (weights,)
    return LinearFn.apply(x, residual, bias, *weights)


# ------------------------------------------------------------------------------------------------- norms / acts
class RMSNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, eps):
        _check_bf16_cuda(x, weight)
        x2 = _as2d(x)
        y, rstd = hip.rmsnorm_fwd(x2, weight, eps)
        ctx.save_for_backward(x2, rstd)
        ctx.weight = weight
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, rstd = ctx.saved_tensors
        w = ctx.weight
        dy2 = _as2d(dy)
        out = {}

        def run(dw_out, acc):
            out["dx"] = hip.rmsnorm_bwd(dy2, x2, w, rstd, dw_out=dw_out, dw_accumulate=acc)

        if ctx.needs_input_grad[1]:
            dw = deliver_vec_grad(w, run)
        else:
            run(None, False)
            dw = None
        return out["dx"].view(dy.shape), dw, None


def rmsnorm(x, weight, eps):
    return RMSNormFn.apply(x, weight, eps)


class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        _check_bf16_cuda(x)
        xc = x.contiguous()
        ctx.save_for_backward(xc)
        ctx.kind = kind
        return hip.act_fwd(xc, kind)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return hip.act_bwd(dy.contiguous(), x, ctx.kind), None


def act(x, kind):
    return ActFn.apply(x, kind)


class LayerNormFn(torch.autograd.Function):
    """Forward-only LayerNorm (the vision tokenizer is frozen in every shipped stage but 'pretrain')."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _check_bf16_cuda(x, weight, bias)
        return hip.layernorm_fwd(_as2d(x), weight, bias, eps).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        raise NotImplementedError("LayerNorm backward: the vision tokenizer is frozen on the SFT/post-training path "
                                  "(models/vlm/prismatic.py:463-467); stage 'pretrain' is not built yet")


def layernorm(x, weight, bias, eps=1e-5):
    return LayerNormFn.apply(x, weight, bias, eps)


class L2NormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _check_bf16_cuda(x)
        y, nrm = hip.l2norm_fwd(_as2d(x))
        ctx.save_for_backward(y, nrm)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        y, nrm = ctx.saved_tensors
        return hip.l2norm_bwd(_as2d(dy), y, nrm).view(dy.shape)


def l2_normalize(x):
    return L2NormFn.apply(x)


class EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, weight):
        _check_bf16_cuda(weight)
        flat = ids.reshape(-1).contiguous()
        ctx.save_for_backward(flat)
        ctx.weight = weight
        return hip.embedding_fwd(flat, weight).view(*ids.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, dy):
        (flat,) = ctx.saved_tensors
        w = ctx.weight
        dy2 = _as2d(dy)
        mg = getattr(w, "main_grad", None)
        if mg is not None:
            if not _touch(w):
                mg.zero_()
            hip.embedding_bwd(flat, dy2, mg)
            return None, None
        g = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
        hip.embedding_bwd(flat, dy2, g)
        return None, g.to(w.dtype)


def embedding(ids, weight):
    return EmbeddingFn.apply(ids, weight)


# ------------------------------------------------------------------------------------------------- losses
class CrossEntropyFn(torch.autograd.Function):
    """mean over rows with label != ignore_index of (logsumexp - logit[label]); logits fp32 or bf16 [rows, V]."""

    @staticmethod
    def forward(ctx, logits2d, labels, ignore_index):
        loss_rows, lse = hip.ce_fwd(logits2d, labels, ignore_index=ignore_index)
        nvalid = (labels != ignore_index).sum().clamp(min=1).to(torch.float32)
        ctx.save_for_backward(logits2d, labels, lse, nvalid)
        ctx.ignore_index = ignore_index
        return loss_rows.sum() / nvalid

    @staticmethod
    def backward(ctx, g):
        logits2d, labels, lse, nvalid = ctx.saved_tensors
        gs = (g.to(torch.float32) / nvalid).reshape(1).contiguous()
        d = hip.ce_bwd(logits2d, labels, lse, gs, 1.0, ignore_index=ctx.ignore_index)
        return d.to(logits2d.dtype), None, None


def cross_entropy(logits2d, labels, ignore_index=-100):
    return CrossEntropyFn.apply(logits2d, labels.contiguous(), ignore_index)


class InfoNCEFn(torch.autograd.Function):
    """Symmetric InfoNCE over M matched rows: logits = a @ b^T / T, loss = (CE(logits, I) + CE(logits^T, I)) / 2
    (models/mla/fuser/contrastive.py:208-215). a, b: [Mp, C] bf16, rows >= M are zero padding (Mp % 128 == 0)."""

    @staticmethod
    def forward(ctx, a, b, M, temperature):
        _check_bf16_cuda(a, b)
        inv_t = 1.0 / temperature
        L = hip.gemm(a, b, out_dtype=torch.float32, alpha=inv_t)
        Lt = hip.gemm(b, a, out_dtype=torch.float32, alpha=inv_t)
        lr, rl = hip.ce_fwd(L[:M], None, ncols=M)
        lc, cl = hip.ce_fwd(Lt[:M], None, ncols=M)
        ctx.save_for_backward(a, b, L, rl, cl)
        ctx.M, ctx.inv_t = M, inv_t
        return (lr.mean() + lc.mean()) * 0.5

    @staticmethod
    def backward(ctx, g):
        a, b, L, rl, cl = ctx.saved_tensors
        gs = g.to(torch.float32).reshape(1).contiguous()
        dL = hip.infonce_bwd(L, rl, cl, gs, ctx.M)
        da = hip.gemm(dL, b, b_mode=1, alpha=ctx.inv_t) if ctx.needs_input_grad[0] else None
        db = hip.gemm(dL, a, a_mode=1, b_mode=1, alpha=ctx.inv_t) if ctx.needs_input_grad[1] else None
        return da, db, None, None


def info_nce(a, b, M, temperature):
    return InfoNCEFn.apply(a, b, M, temperature)


# ------------------------------------------------------------------------------------------------- decoder layer
class DecoderLayerFn(torch.autograd.Function):
    """One whole LlamaDecoderLayer (transformers/models/llama/modeling_llama.py:695-767) as a single autograd node.

    forward : RMSNorm -> fused QKV GEMM -> RoPE (in place) -> causal flash attention -> o_proj GEMM (+residual in
              the epilogue) -> RMSNorm -> fused gate|up GEMM -> SwiGLU -> down GEMM (+residual in the epilogue)
    backward: hand-scheduled; residual-stream gradient adds are fused into the RMSNorm backward kernel, weight
              gradients are written by the wgrad GEMM epilogue into fp32 main_grad.
    save_level: 2 = keep every intermediate; 1 = recompute the two normalised inputs and the SwiGLU product in
                backward (3 HBM-bound kernels); 0 = keep only the layer input and recompute the whole forward
                (activation checkpointing, training/strategies/fsdp.py:211-223).
    """

    @staticmethod
    def _fwd(h2, seqlens, cos, sin, B, S, nheads, eps, w):
        ln1, wq, wk, wv, wo, ln2, wg, wu, wd = w
        H = h2.shape[1]
        D = H // nheads
        xn1, rstd1 = hip.rmsnorm_fwd(h2, ln1, eps)
        qkv = torch.empty((h2.shape[0], 3 * H), dtype=BF16, device=h2.device)
        wqkv = cat_view((wq, wk, wv))
        if wqkv is not None:
            hip.gemm(xn1, wqkv, out=qkv)
        else:
            for i, wi in enumerate((wq, wk, wv)):
                hip.gemm(xn1, wi, out=qkv[:, i * H:(i + 1) * H])
        hip.rope_inplace(qkv, cos, sin, S, nheads, D, 0, H)
        o, lse = hip.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, S, nheads, D, 3 * H, seqlens, 1.0 / math.sqrt(D))
        h1 = hip.gemm(o, wo, residual=h2)
        xn2, rstd2 = hip.rmsnorm_fwd(h1, ln2, eps)
        I = wg.shape[0]
        wgu = cat_view((wg, wu))
        if wgu is not None:
            gu = hip.gemm(xn2, wgu)
        else:
            gu = torch.empty((h2.shape[0], 2 * I), dtype=BF16, device=h2.device)
            hip.gemm(xn2, wg, out=gu[:, :I])
            hip.gemm(xn2, wu, out=gu[:, I:])
        act_ = hip.swiglu_fwd(gu)
        out = hip.gemm(act_, wd, residual=h1)
        return out, (xn1, rstd1, qkv, o, lse, h1, xn2, rstd2, gu, act_)

    @staticmethod
    def forward(ctx, h, seqlens, cos, sin, nheads, eps, save_level, *w):
        _check_bf16_cuda(h, *w)
        B, S, H = h.shape
        h2 = h.reshape(B * S, H)
        if not h2.is_contiguous():
            h2 = h2.contiguous()
        out, (xn1, rstd1, qkv, o, lse, h1, xn2, rstd2, gu, act_) = DecoderLayerFn._fwd(h2, seqlens, cos, sin, B, S, nheads, eps, w)
        ctx.w, ctx.dims, ctx.save_level = w, (B, S, H, nheads, eps), save_level
        ctx.aux = (seqlens, cos, sin)
        if save_level >= 2:
            ctx.save_for_backward(h2, xn1, rstd1, qkv, o, lse, h1, xn2, rstd2, gu, act_)
        elif save_level == 1:
            ctx.save_for_backward(h2, rstd1, qkv, o, lse, h1, rstd2, gu)
        else:
            ctx.save_for_backward(h2)
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, dout):
        """All-NT backward: every large GEMM gets k-contiguous operands (W^T, x^T, dy^T from the HBM-bound tile-transpose
        kernels; x^T of the normalised inputs and of the SwiGLU product are recomputed straight into transposed layout),
        so dgrad and wgrad run on the same 256x256 ds_read_b128 kernel as the forward (mla_amd/csrc/transpose.hip)."""
        w = ctx.w
        ln1, wq, wk, wv, wo, ln2, wg, wu, wd = w
        B, S, H, nheads, eps = ctx.dims
        seqlens, cos, sin = ctx.aux
        D = H // nheads
        T = B * S
        lvl = ctx.save_level
        xn1 = xn2 = act_ = None
        if lvl >= 2:
            h2, xn1, rstd1, qkv, o, lse, h1, xn2, rstd2, gu, act_ = ctx.saved_tensors
        elif lvl == 1:
            h2, rstd1, qkv, o, lse, h1, rstd2, gu = ctx.saved_tensors
        else:
            (h2,) = ctx.saved_tensors
            _, (xn1, rstd1, qkv, o, lse, h1, xn2, rstd2, gu, act_) = DecoderLayerFn._fwd(h2, seqlens, cos, sin, B, S, nheads, eps, w)
        if T % 8 != 0:
            raise NotImplementedError("decoder backward needs batch*seq to be a multiple of 8 (tile transposes)")
        need = ctx.needs_input_grad[7:]
        d2 = dout.reshape(T, H)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        grads: List[Optional[torch.Tensor]] = [None] * 9

        def wT(ws):
            wc = cat_view(ws)
            return hip.transpose(wc if wc is not None else torch.cat(list(ws), 0))

        # ---- MLP: down projection
        dact = hip.gemm(d2, wT((wd,)))                                   # [T, I]
        if need[8]:
            actT = hip.swiglu_fwd_t(gu) if act_ is None else hip.transpose(act_)
            grads[8] = deliver_wgrad_nt((wd,), hip.transpose(d2), actT, need[8:9])[0]
            del actT
        act_ = None
        dgu, _ = hip.swiglu_bwd(dact, gu)
        del dact
        # ---- MLP: gate | up projection
        dxn2 = hip.gemm(dgu, wT((wg, wu)))                               # [T, H], K = 2I
        if need[6] or need[7]:
            xn2T = hip.rmsnorm_apply_t(h1, ln2, rstd2) if xn2 is None else hip.transpose(xn2)
            grads[6], grads[7] = deliver_wgrad_nt((wg, wu), hip.transpose(dgu), xn2T, need[6:8])
            del xn2T
        del dgu
        xn2 = None
        holder = {}

        def ln2_run(dw_out, acc):
            holder["dh1"] = hip.rmsnorm_bwd(dxn2, h1, ln2, rstd2, dres=d2, dw_out=dw_out, dw_accumulate=acc)

        if need[5]:
            grads[5] = deliver_vec_grad(ln2, ln2_run)
        else:
            ln2_run(None, False)
        dh1 = holder["dh1"]
        del dxn2

        # ---- attention output projection
        do = hip.gemm(dh1, wT((wo,)))
        if need[4]:
            grads[4] = deliver_wgrad_nt((wo,), hip.transpose(dh1), hip.transpose(o), need[4:5])[0]
        dqkv = torch.empty_like(qkv)
        hip.attn_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], o, do, lse, seqlens, dqkv[:, :H], dqkv[:, H:2 * H],
                     dqkv[:, 2 * H:], B, S, nheads, D, 3 * H, 1.0 / math.sqrt(D))
        del do
        hip.rope_inplace(dqkv, cos, sin, S, nheads, D, 0, H, backward=True)
        # ---- q | k | v projection
        dxn1 = hip.gemm(dqkv, wT((wq, wk, wv)))                          # [T, H], K = 3H
        if need[1] or need[2] or need[3]:
            xn1T = hip.rmsnorm_apply_t(h2, ln1, rstd1) if xn1 is None else hip.transpose(xn1)
            grads[1], grads[2], grads[3] = deliver_wgrad_nt((wq, wk, wv), hip.transpose(dqkv), xn1T, need[1:4])
            del xn1T
        del dqkv

        def ln1_run(dw_out, acc):
            holder["dh"] = hip.rmsnorm_bwd(dxn1, h2, ln1, rstd1, dres=dh1, dw_out=dw_out, dw_accumulate=acc)

        if need[0]:
            grads[0] = deliver_vec_grad(ln1, ln1_run)
        else:
            ln1_run(None, False)
        dh = holder["dh"].view(B, S, H) if ctx.needs_input_grad[0] else None
        return (dh, None, None, None, None, None, None, *grads)


def decoder_layer(h, seqlens, cos, sin, nheads, eps, save_level, weights):
    return DecoderLayerFn.apply(h, seqlens, cos, sin, nheads, eps, save_level, *weights)


class GatherRowsFn(torch.autograd.Function):
    """out[r] = src[idx[r]] over rows of a 2-D bf16 tensor; idx must be injective (backward is a plain scatter)."""

    @staticmethod
    def forward(ctx, src2d, idx):
        _check_bf16_cuda(src2d)
        ctx.save_for_backward(idx)
        ctx.n_src = src2d.shape[0]
        return hip.gather_rows(src2d.contiguous(), idx)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return hip.gather_rows(dy.contiguous(), idx, out_rows=ctx.n_src, scatter=True), None


def gather_rows(src2d, idx):
    return GatherRowsFn.apply(src2d, idx.contiguous())


class UnitBoundaryFn(torch.autograd.Function):
    """Identity whose backward fires ``hook()``: placed on a unit's input, it runs once every gradient kernel of that
    unit has been enqueued -- mla_amd.fsdp uses it to start the unit's gradient reduce-scatter on the side stream."""

    @staticmethod
    def forward(ctx, x, hook):
        ctx.hook = hook
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.hook()
        return g, None


def unit_boundary(x, hook):
    return UnitBoundaryFn.apply(x, hook) if hook is not None else x
