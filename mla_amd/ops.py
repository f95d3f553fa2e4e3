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
import threading
import os
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


def _is_touched(w) -> bool:
    """Whether main_grad (of a parameter, or of the region a param_view covers) already holds a contribution from this
    accumulation window. The state of a view lives on its parent parameter: views are re-created every forward."""
    region = getattr(w, "_mg_region", None)
    if region is not None:
        return bool(region[0]._mg_regions.get(region[1], False))
    return bool(getattr(w, "_mg_touched", False))


def _assert_unlocked(*weights) -> None:
    """Raises BEFORE anything is enqueued into main_grad of a parameter whose sharding unit's reduce-scatter has been launched in this
    step (mla_amd/fsdp.py: zero_stale_and_lock): its gradient buffer is being read by the collective, or already holds the reduced
    shard in place -- e.g. the same decoder layer run through backward twice in one step. (Advisor, round 5: the check used to sit only
    in _mark_touched, i.e. AFTER the wgrad GEMM into the buffer had been queued.)"""
    for w in weights:
        region = getattr(w, "_mg_region", None)
        owner = region[0] if region is not None else w
        if getattr(owner, "_mg_locked", False):
            raise RuntimeError("main_grad written after its sharding unit's reduce-scatter was launched in this step")


def _mark_touched(w) -> None:
    region = getattr(w, "_mg_region", None)
    _assert_unlocked(w)                                  # backstop: every writer checks before its launch
    if region is not None:
        region[0]._mg_regions[region[1]] = True
    else:
        w._mg_touched = True


def _touch(param) -> bool:
    """Returns whether main_grad already holds a contribution from this window (then accumulate) and marks it."""
    acc = _is_touched(param)
    _mark_touched(param)
    return acc


def _sq_stale(*weights) -> None:
    """main_grad of these parameters (or of the parents of param_views) is about to be written by a launch that leaves no
    sum(dW^2) partials: the owner drops every recorded gradient-norm range that intersects them (FlatUnit.sq_invalidate), so the
    clipping norm re-reads those elements instead of trusting partials of an earlier micro-batch."""
    for w in weights:
        region = getattr(w, "_mg_region", None)
        p = region[0] if region is not None else w
        inv = getattr(p, "_sq_invalidate", None)
        if inv is not None:
            inv(p)


def reset_main_grad_state(params) -> None:
    """Start a new accumulation window for parameters that carry ``main_grad`` but are NOT owned by a FlatUnit (ad-hoc / unit-test
    use; FlatUnit.begin_step does this for its own): the next backward overwrites main_grad instead of adding to it."""
    for p in params:
        if hasattr(p, "_mg_touched"):
            p._mg_touched = False
        if hasattr(p, "_mg_regions"):
            p._mg_regions = {}


def deliver_wgrad(weights: Sequence[torch.Tensor], dy2: torch.Tensor, x2: torch.Tensor, needs: Sequence[bool]):
    """dW_i = dy[:, slice_i]^T @ x for every weight; into main_grad when present, else returned as tensors."""
    grads: List[Optional[torch.Tensor]] = [None] * len(weights)
    mgs = [getattr(w, "main_grad", None) for w in weights]
    _assert_unlocked(*[w for w, m, n in zip(weights, mgs, needs) if m is not None and n])
    if all(m is not None for m in mgs) and all(needs):
        mcat = cat_view(mgs)
        states = {_is_touched(w) for w in weights}
        if mcat is not None and len(states) == 1:
            acc = states.pop()
            _sq_stale(*weights)
            hip.gemm(dy2, x2, out=mcat, a_mode=1, b_mode=1, accumulate=acc)
            for w in weights:
                _mark_touched(w)
            return grads
    off = 0
    for i, w in enumerate(weights):
        n = w.shape[0]
        if needs[i]:
            dys = dy2[:, off:off + n]
            if mgs[i] is not None:
                _sq_stale(w)
                hip.gemm(dys, x2, out=mgs[i], a_mode=1, b_mode=1, M=n, accumulate=_touch(w))
            else:
                grads[i] = hip.gemm(dys, x2, a_mode=1, b_mode=1, M=n, out_dtype=torch.float32).to(w.dtype)
        off += n
    return grads


_WGRAD_SQ = os.environ.get("MLA_WGRAD_SQ", "1") != "0"      # A/B switch (tools): gradient-norm partials from the wgrad epilogues


def _gemm_into_main_grad(weights, dyT, xT, out, accumulate: bool) -> None:
    """dW (+)= dyT xT^T into the fp32 gradient buffer. When the owner of the buffer collects the clipping norm itself (one process,
    no reduce-scatter: FlatUnit.sq_sink) the same launch also leaves sum(dW^2) of the final values as partial sums, so the norm
    never re-reads these 4 bytes per parameter (training/strategies/fsdp.py:308-310)."""
    _assert_unlocked(*weights)
    sink = getattr(weights[0], "_sq_sink", None) if _WGRAD_SQ else None
    owner = getattr(sink, "__self__", None)      # (a bound method is a fresh object per access: compare the owners)
    if sink is not None and all(getattr(getattr(w, "_sq_sink", None), "__self__", None) is owner and
                                getattr(w, "_mg_region", None) is None for w in weights):
        need = hip.gemm_sq_slots(dyT, xT, out)
        if need > 0:
            part = sink(weights, need)                       # `need` floats of the owner's partial-sum arena (None: arena full)
            if part is not None:
                hip.gemm_sq(dyT, xT, out, accumulate, part)
                return
    # no partials from this launch (odd shape, arena full, param_view region, foreign owner): whatever an earlier micro-batch of this
    # window recorded for the range is stale once this launch has added to / overwritten it
    _sq_stale(*weights)
    hip.gemm(dyT, xT, out=out, accumulate=accumulate)


def deliver_wgrad_nt(weights: Sequence[torch.Tensor], dyT: torch.Tensor, xT: torch.Tensor, needs: Sequence[bool]):
    """Same as deliver_wgrad with pre-transposed operands: dW_i = dyT[rows_i] @ xT^T  (dyT [sum N_i, T], xT [K, T], both
    k-contiguous -> the fast NT kernel)."""
    grads: List[Optional[torch.Tensor]] = [None] * len(weights)
    mgs = [getattr(w, "main_grad", None) for w in weights]
    _assert_unlocked(*[w for w, m, n in zip(weights, mgs, needs) if m is not None and n])
    if all(m is not None for m in mgs) and all(needs):
        mcat = cat_view(mgs)
        states = {_is_touched(w) for w in weights}
        if mcat is not None and len(states) == 1:
            _gemm_into_main_grad(weights, dyT, xT, mcat, states.pop())
            for w in weights:
                _mark_touched(w)
            return grads
    off = 0
    for i, w in enumerate(weights):
        n = w.shape[0]
        if needs[i]:
            if mgs[i] is not None:
                _gemm_into_main_grad((w,), dyT[off:off + n], xT, mgs[i], _touch(w))
            else:
                grads[i] = hip.gemm(dyT[off:off + n], xT, out_dtype=torch.float32).to(w.dtype)
        off += n
    return grads


def deliver_vec_grad(param: torch.Tensor, compute):
    """compute(out_f32, accumulate) fills a 1-D fp32 gradient. Routes to main_grad or returns a tensor."""
    mg = getattr(param, "main_grad", None)
    if mg is not None:
        _sq_stale(param)
        compute(mg, _touch(param))
        return None
    g = torch.empty(param.shape, dtype=torch.float32, device=param.device)
    compute(g, False)
    return g.to(param.dtype)


def _pad8(n: int) -> int:
    return (n + 7) // 8 * 8


def _pad_tokens(n: int) -> int:
    """Token rows of a decoder layer: a multiple of 8 (the tile transposes of the all-NT backward move 8 tokens per 16-B access) and, for
    more than a tile of tokens, of 64 -- the token count is the REDUCTION dimension of every wgrad GEMM: K % 32 != 0 sends a GEMM to
    the SIMT fallback kernel (round 6: the shared-prefix layout has 8 x 2 057 = 16 456 rows; unpadded its step ran at 118 TFLOP/s), and the
    256-tile kernel's K step is 64. The benchmark shapes (17 536 = 274 x 64, 65 536) are multiples of 64 already."""
    return (n + 63) // 64 * 64 if n > 256 else (n + 7) // 8 * 8


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
        T, N = dy2.shape
        K = x2.shape[1]
        # large aligned shapes: transpose the operands once and stay on the k-contiguous (NT) MFMA kernel, which is ~2x faster
        # than the reduction-major variants (DESIGN.md "all-NT backward")
        big = T >= 256 and N >= 256 and K >= 256 and T % 64 == 0 and N % 8 == 0 and K % 64 == 0 and N % 64 == 0
        wcat = cat_view(weights) if big else None
        if wcat is not None and dy2.stride(0) == N and x2.stride(0) == K:
            if ni[0]:
                dx = hip.gemm(dy2, hip.transpose(wcat)).view(ctx.in_shape)
            wg = [None] * len(weights)
            if any(ni[3:]):
                wg = deliver_wgrad_nt(weights, hip.transpose(dy2), hip.transpose(x2), ni[3:])
            db = None
            if ctx.bias is not None and ni[2]:
                db = deliver_vec_grad(ctx.bias, lambda out, acc: hip.colsum(dy2, out, acc))
            return (dx, dy if (ctx.has_res and ni[1]) else None, db, *wg)
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


class TimmRmsNormFn(torch.autograd.Function):
    """timm==0.9.10 RmsNorm: x * rsqrt(torch.var(x, -1) + eps) * weight (hip.timm_rmsnorm_fwd)."""

    @staticmethod
    def forward(ctx, x, weight, eps):
        _check_bf16_cuda(x, weight)
        x2 = _as2d(x)
        y, mean, rstd = hip.timm_rmsnorm_fwd(x2, weight, eps)
        ctx.save_for_backward(x2, mean, rstd)
        ctx.weight = weight
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        w = ctx.weight
        dy2 = _as2d(dy)
        out = {}

        def run(dw_out, acc):
            out["dx"] = hip.timm_rmsnorm_bwd(dy2, x2, w, mean, rstd, dw_out=dw_out, dw_accumulate=acc)

        if ctx.needs_input_grad[1]:
            dw = deliver_vec_grad(w, run)
        else:
            run(None, False)
            dw = None
        return out["dx"].view(dy.shape), dw, None


def timm_rmsnorm(x, weight, eps):
    return TimmRmsNormFn.apply(x, weight, eps)


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


def _deliver_small(param, g32):
    """Route a small fp32 gradient (already reduced) to main_grad or hand it back to autograd."""
    mg = getattr(param, "main_grad", None)
    if mg is None:
        return g32.to(param.dtype).view(param.shape)
    _sq_stale(param)
    if _touch(param):
        mg.add_(g32.view(mg.shape))
    else:
        mg.copy_(g32.view(mg.shape))
    return None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm (fp32 statistics). Backward: models/mla/generation/models.py:43-46,126 (trainable heads)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _check_bf16_cuda(x, weight, bias)
        x2 = _as2d(x)
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            return hip.layernorm_fwd(x2, weight, bias, eps).view(x.shape)
        y, mean, rstd = hip.layernorm_stats_fwd(x2, weight, bias, eps)
        ctx.save_for_backward(x2, mean, rstd)
        ctx.weight, ctx.bias = weight, bias
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        H = x2.shape[1]
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dwb = torch.empty((2, H), dtype=torch.float32, device=x2.device) if need_w else None
        dx = hip.layernorm_bwd(_as2d(dy), x2, ctx.weight, mean, rstd, dw=dwb[0] if need_w else None, db=dwb[1] if need_w else None)
        dw = _deliver_small(ctx.weight, dwb[0]) if ctx.needs_input_grad[1] else None
        db = _deliver_small(ctx.bias, dwb[1]) if ctx.needs_input_grad[2] else None
        return dx.view(dy.shape), dw, db, None


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
            _sq_stale(w)
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
_ROPE_EPILOGUE = os.environ.get("MLA_ROPE_EPILOGUE", "1") != "0"   # A/B switch (tools): 0 = separate RoPE pass after the QKV GEMM
_SWIGLU_FWD_EPILOGUE = os.environ.get("MLA_SWIGLU_FWD_EPILOGUE", "1") != "0"   # A/B switch: 0 = gate|up GEMM + separate SwiGLU pass
_SWIGLU_BWD_EPILOGUE = os.environ.get("MLA_SWIGLU_BWD_EPILOGUE", "1") != "0"   # A/B switch: 0 = d(act) GEMM + separate SwiGLU backward
_ATTN_BWD_T = os.environ.get("MLA_ATTN_BWD_T", "1") != "0"           # dqkv^T / o^T written by the attention-backward kernels
_SWIGLU_DUAL = os.environ.get("MLA_SWIGLU_DUAL", "1") != "0"     # A/B switch (tools): 0 = recompute act^T in the backward
_RECOMPUTE_LEAN = os.environ.get("MLA_RECOMPUTE_LEAN", "1") != "0"   # A/B switch: 0 = a checkpointed layer recomputes its WHOLE forward (incl. the unused down projection)


# RMSNorm folded into the projections (round 6, mla_hip.h: mla_gemm_res_norm / _qkv_rope_rs / _gateup_swiglu_rs): the stand-alone norm passes
# of the decoder layer's FORWARD disappear -- the GEMM that produces the residual-stream rows leaves x * g and the partials of sum(x^2), the
# projection applies rstd to its accumulator rows. Opt-in: MLA_NORM_FOLD=1 (or set_norm_fold(True)).
# Built, bit-exact against the plain kernels, closer to fp32 than the separate norm (7B layer: 0.63-0.71 x the reference's own bf16 error) --
# and OFF by default: six alternating same-box pairs of the configs[1] step measured +2.4 ms (95 % CI [+1.2, +3.6]) WITH it
# (profiles/r6_norm_fold_ab.txt). Per launch at T = 17 536: the norm pass it removes is 48.5 us (reads and writes at 5.9 TB/s); what it adds
# is the x * g store inside the producers' epilogue bursts (+39 / +47 us, all 256 CUs storing at once) and the partials -> rstd prologue of
# every consumer tile (+25 / +59 us); history/round_6.md section 9.
_NORM_FOLD = os.environ.get("MLA_NORM_FOLD", "0") != "0"


def norm_fold_enabled() -> bool:
    return _NORM_FOLD


def set_norm_fold(on: bool) -> bool:
    """A/B switch for tests and tools; returns the previous setting."""
    global _NORM_FOLD
    prev, _NORM_FOLD = _NORM_FOLD, bool(on)
    return prev


class NormFoldIO:
    """What one decoder layer hands to the next when the RMSNorms are folded into the projections: `pre` = (rows' data_ptr, norm weight,
    xg = bf16(h * g), ss partials) made by the PREVIOUS layer's down projection for THIS layer's input_layernorm (None: first layer, or
    the previous layer could not make it -> mla_rmsnorm_prep); `next_ln` = the next layer's input_layernorm weight (None: last layer);
    `out` = what this layer's down projection made for the next one."""
    __slots__ = ("pre", "next_ln", "out")

    def __init__(self, pre=None, next_ln=None):
        self.pre, self.next_ln, self.out = pre, next_ln, None


# Suffix groups of shared-prefix sequences (round 6, mla_hip.h: mla_attn_fwd_g): (first suffix row, rows per group) or None. Set by the
# caller of the decoder stack for the duration of its forward (LlamaModel.forward(attn_groups=...)); every DecoderLayerFn.forward
# records the value it saw in its ctx, so backward and checkpoint recomputation use the same grouping whatever is current then.
_ATTN_GROUPS = threading.local()


def current_attn_groups():
    return getattr(_ATTN_GROUPS, "value", None)


class attn_groups:
    """with ops.attn_groups((start, length)): ... -- the decoder layers run inside see the suffix-group attention mask."""

    def __init__(self, groups):
        self.groups = groups

    def __enter__(self):
        self.prev = current_attn_groups()
        _ATTN_GROUPS.value = self.groups

    def __exit__(self, *a):
        _ATTN_GROUPS.value = self.prev


class DecoderLayerFn(torch.autograd.Function):
    """One whole LlamaDecoderLayer (transformers/models/llama/modeling_llama.py:695-767) as a single autograd node.

    forward : RMSNorm -> fused QKV GEMM -> RoPE (in place) -> causal flash attention -> o_proj GEMM (+residual in
              the epilogue) -> RMSNorm -> fused gate|up GEMM -> SwiGLU -> down GEMM (+residual in the epilogue)
    backward: hand-scheduled; residual-stream gradient adds are fused into the RMSNorm backward kernel, weight
              gradients are written by the wgrad GEMM epilogue into fp32 main_grad.
    save_level: 2 = keep every intermediate; 1 = recompute the two normalised inputs in the backward (HBM-bound passes) and keep
                the SwiGLU product transposed; 3 = level 1 WITHOUT the kept product (it is recomputed from gate|up, one more
                HBM-bound pass; 19 % less memory per layer -- the filler level of a mixed policy); 0 = keep only the layer input
                and recompute the whole forward (activation checkpointing, training/strategies/fsdp.py:211-223).
                Every level runs the same kernels on the same inputs: outputs and gradients are bit-identical across levels.
    """

    @staticmethod
    def _fold_ok(h2, cos, sin, Sr, nheads, w, save_t) -> bool:
        """The folded-RMSNorm forward needs all four GEMMs of the layer inside the 256x256 kernel's fused-epilogue contracts."""
        ln1, wq, wk, wv, wo, ln2, wg, wu, wd = w
        T, H = h2.shape
        wqkv, wgu = cat_view((wq, wk, wv)), cat_view((wg, wu))
        if wqkv is None or wgu is None or H // nheads != 128 or not (_ROPE_EPILOGUE and _SWIGLU_FWD_EPILOGUE):
            return False
        if T < 256 or H % 256 != 0 or any(t.data_ptr() % 16 for t in (ln1, ln2, wo, wd)) or wo.stride(0) % 8 or wd.stride(0) % 8:
            return False
        qkv_like = torch.empty((0, 3 * H), dtype=BF16, device=h2.device)
        return (hip.qkv_rope_ok(h2, wqkv, qkv_like, cos, sin, Sr, 2 * H) and hip.gateup_swiglu_ok(h2, wgu, save_t and T % 8 == 0) and
                wd.shape[1] % 64 == 0)

    @staticmethod
    def _fwd_folded(h2, seqlens, cos, sin, B, S, nheads, eps, w, save_t, groups, fold, need_out=True, need_gu=True):
        """_fwd with both RMSNorms folded into the projections (mla_hip.h "RMSNorm folded into the projections"): no stand-alone norm
        pass; rstd1 / rstd2 come out of the QKV and gate|up launches. fold = NormFoldIO, or a saved rstd1 tensor (recomputation of a
        checkpointed layer: the same row scale as in the forward, whatever produced it there)."""
        ln1, wq, wk, wv, wo, ln2, wg, wu, wd = w
        T, H = h2.shape
        D = H // nheads
        Sr = cos.shape[0]
        wqkv, wgu = cat_view((wq, wk, wv)), cat_view((wg, wu))
        io = fold if isinstance(fold, NormFoldIO) else None
        pre = io.pre if io is not None else None
        if pre is not None and not (pre[0] == h2.data_ptr() and pre[1].data_ptr() == ln1.data_ptr() and pre[2].shape == h2.shape):
            pre = None                                   # made for other rows / another weight: ignore it
        if pre is not None:
            xg1, norm1 = pre[2], (pre[3], None, eps)
        elif io is None and fold is not None:
            xg1, norm1 = hip.rmsnorm_prep(h2, ln1, eps, want_rstd=False)[0], (None, fold, eps)
        else:
            xg1, r1 = hip.rmsnorm_prep(h2, ln1, eps)
            norm1 = (None, r1, eps)
        qkv = torch.empty((T, 3 * H), dtype=BF16, device=h2.device)
        rstd1 = hip.gemm_qkv_rope(xg1, wqkv, qkv, cos, sin, Sr, 2 * H, norm=norm1)      # fused RMSNorm + QKV + RoPE
        assert rstd1 is not False
        del xg1
        o, lse = hip.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, S, nheads, D, 3 * H, seqlens, 1.0 / math.sqrt(D),
                              rows=T, groups=groups)
        h1, xg2, ss2 = hip.gemm_res_norm(o, wo, h2, ln2)
        gu, act_, actT, rstd2 = hip.gemm_gateup_swiglu(xg2, wgu, save_t and T % 8 == 0, norm=(ss2, None, eps), want_act=need_out, want_gu=need_gu)
        del xg2, ss2
        if not need_out:
            out = None
        elif io is not None and io.next_ln is not None and io.next_ln.data_ptr() % 16 == 0:
            out, xgn, ssn = hip.gemm_res_norm(act_, wd, h1, io.next_ln)
            io.out = (out.data_ptr(), io.next_ln, xgn, ssn)
        else:
            out = hip.gemm(act_, wd, residual=h1)
        return out, (None, rstd1, qkv, o, lse, h1, None, rstd2, gu, act_, actT)

    @staticmethod
    def _fwd(h2, seqlens, cos, sin, B, S, nheads, eps, w, save_t=False, groups=None, fold=None, need_out=True, need_gu=True):
        """need_out=False: the recomputation of a checkpointed layer inside its backward -- everything up to the SwiGLU product, NOT the
        down projection (its output is the layer output, which the backward never reads; torch.utils.checkpoint, the reference's
        fsdp.py:211-223, recomputes it anyway: 22 % of a layer's forward FLOPs). need_gu=False: the FORWARD of a checkpointed layer --
        gate|up never leaves the chip (only the SwiGLU product does; nothing but the layer input is kept)."""
        ln1, wq, wk, wv, wo, ln2, wg, wu, wd = w
        H = h2.shape[1]
        D = H // nheads
        if fold is not None:
            return DecoderLayerFn._fwd_folded(h2, seqlens, cos, sin, B, S, nheads, eps, w, save_t, groups, fold, need_out, need_gu)
        xn1, rstd1 = hip.rmsnorm_fwd(h2, ln1, eps)
        qkv = torch.empty((h2.shape[0], 3 * H), dtype=BF16, device=h2.device)
        wqkv = cat_view((wq, wk, wv))
        # fused RMSNorm-output x [Wq|Wk|Wv]^T + RoPE: the rotary embedding of q and k happens in the GEMM epilogue (north_star's
        # "fused RoPE + QKV"); shapes outside the fused kernel's contract take the two separate launches
        # (tables with B * S rows = per-sample positions, shared-prefix sequences of ragged prompts: position = row % (B * S) = the row)
        Sr = cos.shape[0]
        assert Sr == S or (Sr == B * S and groups is not None), (Sr, S, B)
        if not (wqkv is not None and D == 128 and _ROPE_EPILOGUE and
                hip.gemm_qkv_rope(xn1, wqkv, qkv, cos, sin, Sr, 2 * H)):
            if wqkv is not None:
                hip.gemm(xn1, wqkv, out=qkv)
            else:
                for i, wi in enumerate((wq, wk, wv)):
                    hip.gemm(xn1, wi, out=qkv[:, i * H:(i + 1) * H])
            hip.rope_inplace(qkv, cos, sin, Sr, nheads, D, 0, H)
        o, lse = hip.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, S, nheads, D, 3 * H, seqlens, 1.0 / math.sqrt(D),
                              rows=h2.shape[0], groups=groups)
        h1 = hip.gemm(o, wo, residual=h2)
        xn2, rstd2 = hip.rmsnorm_fwd(h1, ln2, eps)
        I = wg.shape[0]
        wgu = cat_view((wg, wu))
        # fused gate|up projection + SwiGLU: the product (and, with save_t, its transposed copy for the backward's wgrad) is formed in the
        # GEMM epilogue -- gu is written once and not read again in the forward pass
        fused = (hip.gemm_gateup_swiglu(xn2, wgu, save_t and h2.shape[0] % 8 == 0, want_act=need_out, want_gu=need_gu)
                 if (wgu is not None and _SWIGLU_FWD_EPILOGUE) else None)
        if fused is not None:
            gu, act_, actT = fused
        else:
            if wgu is not None:
                gu = hip.gemm(xn2, wgu)
            else:
                gu = torch.empty((h2.shape[0], 2 * I), dtype=BF16, device=h2.device)
                hip.gemm(xn2, wg, out=gu[:, :I])
                hip.gemm(xn2, wu, out=gu[:, I:])
            # save_t: the caller keeps the SwiGLU product for the backward in TRANSPOSED layout (the wgrad operand), written by the
            # same pass that produces the row-major copy for the down projection
            if save_t and gu.shape[0] % 8 == 0:
                act_, actT = hip.swiglu_fwd_dual(gu)
            else:
                act_, actT = hip.swiglu_fwd(gu), None
        out = hip.gemm(act_, wd, residual=h1) if need_out else None
        return out, (xn1, rstd1, qkv, o, lse, h1, xn2, rstd2, gu, act_, actT)

    @staticmethod
    def forward(ctx, h, seqlens, cos, sin, nheads, eps, save_level, fold_io, *w):
        _check_bf16_cuda(h, *w)
        B, S, H = h.shape
        h2 = h.reshape(B * S, H)
        if not h2.is_contiguous():
            h2 = h2.contiguous()
        T, Tp = B * S, _pad_tokens(B * S)
        if Tp != T:
            # the tile transposes of the all-NT backward move 8 tokens per 16-B access: an odd token count (per-device batch 1 with
            # an odd padded length) runs on zero rows appended here; they stay zero through every row-wise op, contribute zero to
            # every weight gradient, and are cut off again below
            h2 = torch.cat([h2, h2.new_zeros(Tp - T, H)], 0)
        keep_t = save_level == 1 and ctx.needs_input_grad[8 + 8] and _SWIGLU_DUAL   # down_proj trainable: its wgrad wants act^T
        if save_level == 3:
            save_level = 1                         # "1-lean": same saved set as level 1 minus act^T
        ctx.groups = current_attn_groups()
        # RMSNorms folded into the projections: when the caller hands a NormFoldIO and every GEMM of the layer is inside the fused kernels'
        # contracts. The padded rows of a handed-over x * g belong to the previous layer's padded output: same zero rows.
        fold = fold_io if (fold_io is not None and DecoderLayerFn._fold_ok(h2, cos, sin, cos.shape[0], nheads, w, keep_t)) else None
        if fold is not None and fold.pre is not None and Tp != T and fold.pre[0] == h.data_ptr() and fold.pre[2].shape[0] == Tp:
            fold.pre = (h2.data_ptr(),) + tuple(fold.pre[1:])
        ctx.folded = fold is not None
        out, (xn1, rstd1, qkv, o, lse, h1, xn2, rstd2, gu, act_, actT) = DecoderLayerFn._fwd(
            h2, seqlens, cos, sin, B, S, nheads, eps, w, save_t=keep_t, groups=ctx.groups, fold=fold,
            need_gu=not (save_level == 0 and _RECOMPUTE_LEAN))
        out = out[:T]
        ctx.w, ctx.dims, ctx.save_level = w, (B, S, H, nheads, eps), save_level
        ctx.aux = (seqlens, cos, sin)
        ctx.has_actT = actT is not None
        if save_level >= 2:
            ctx.save_for_backward(h2, xn1, rstd1, qkv, o, lse, h1, xn2, rstd2, gu, act_)
        elif save_level == 1:
            # level 1 = recompute the two normalised inputs in the backward (one HBM-bound pass each, written straight into the
            # transposed layout) and KEEP the SwiGLU product, already transposed (+0.39 GB per layer at 7B; 288 GB of HBM)
            ctx.save_for_backward(*((h2, rstd1, qkv, o, lse, h1, rstd2, gu) + ((actT,) if actT is not None else ())))
        elif ctx.folded:
            ctx.save_for_backward(h2, rstd1)       # the recomputation scales the QKV rows by the SAME rstd the forward used
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
        xn1 = xn2 = act_ = actT_saved = None
        if lvl >= 2:
            h2, xn1, rstd1, qkv, o, lse, h1, xn2, rstd2, gu, act_ = ctx.saved_tensors
        elif lvl == 1:
            if ctx.has_actT:
                h2, rstd1, qkv, o, lse, h1, rstd2, gu, actT_saved = ctx.saved_tensors
            else:
                h2, rstd1, qkv, o, lse, h1, rstd2, gu = ctx.saved_tensors
        else:
            if ctx.folded:
                h2, rstd1_f = ctx.saved_tensors
            else:
                (h2,), rstd1_f = ctx.saved_tensors, None
            # the recomputation stops in front of the down projection and asks the gate|up epilogue for the SwiGLU product in the
            # layout the down-projection wgrad wants (round 6: was a full forward + a transpose pass over act)
            want_t = ctx.needs_input_grad[8 + 8] and _SWIGLU_DUAL and _RECOMPUTE_LEAN
            _, (xn1, rstd1, qkv, o, lse, h1, xn2, rstd2, gu, act_, actT_saved) = DecoderLayerFn._fwd(
                h2, seqlens, cos, sin, B, S, nheads, eps, w, save_t=want_t, groups=ctx.groups, fold=rstd1_f, need_out=not _RECOMPUTE_LEAN)
        need = ctx.needs_input_grad[8:]
        d2 = dout.reshape(T, H)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        Tr, T = T, _pad_tokens(T)               # row padding of forward(): zero gradient rows
        if T != Tr:
            d2 = torch.cat([d2, d2.new_zeros(T - Tr, H)], 0)
        grads: List[Optional[torch.Tensor]] = [None] * 9

        def wT(ws):
            wc = cat_view(ws)
            return hip.transpose(wc if wc is not None else torch.cat(list(ws), 0))

        # ---- MLP: down projection
        want_w = need[6] or need[7]
        wdT = wT((wd,))
        fused = hip.gemm_dact_swiglu_bwd(d2, wdT, gu) if (want_w and _SWIGLU_BWD_EPILOGUE) else None   # d(act) never leaves the chip
        dact = hip.gemm(d2, wdT) if fused is None else None              # [T, I]
        del wdT
        if need[8]:
            actT = actT_saved if actT_saved is not None else (hip.swiglu_fwd_t(gu) if act_ is None else hip.transpose(act_))
            grads[8] = deliver_wgrad_nt((wd,), hip.transpose(d2), actT, need[8:9])[0]
            del actT
        act_ = actT_saved = None
        fuse_t = want_w and gu.shape[0] % 8 == 0         # dgu and dgu^T from one pass (saves re-reading the 2I-wide gradient)
        if fused is not None:
            dgu, dguT = fused
        elif fuse_t:
            dgu, dguT = hip.swiglu_bwd_t(dact, gu)
        else:
            dgu, _ = hip.swiglu_bwd(dact, gu)
        del dact, fused
        # ---- MLP: gate | up projection
        dxn2 = hip.gemm(dgu, wT((wg, wu)))                               # [T, H], K = 2I
        if want_w:
            xn2T = hip.rmsnorm_apply_t(h1, ln2, rstd2) if xn2 is None else hip.transpose(xn2)
            grads[6], grads[7] = deliver_wgrad_nt((wg, wu), dguT if fuse_t else hip.transpose(dgu), xn2T, need[6:8])
            del xn2T
            dguT = None
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
        dqkv = torch.empty_like(qkv)
        if T != Tr:
            dqkv[Tr:].zero_()
        # the RoPE backward of dq / dk is applied in the attention-backward epilogues (no separate in-place pass over dqkv)
        fuse_rope = cos.shape[0] in (S, B * S) and cos.is_contiguous() and sin.is_contiguous() and cos.dtype == torch.float32
        want_qkv_w = need[1] or need[2] or need[3]
        # dqkv^T and o^T (wgrad operands) leave the attention-backward kernels with the rows: no transpose passes over them
        tr = None
        if _ATTN_BWD_T and fuse_rope and S % 4 == 0 and want_qkv_w and need[4]:
            tr = (torch.empty((3 * H, T), dtype=BF16, device=qkv.device), torch.empty((H, T), dtype=BF16, device=qkv.device))
            if T != Tr:
                tr[0][:, Tr:].zero_()
                tr[1][:, Tr:].zero_()
        hip.attn_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], o, do, lse, seqlens, dqkv[:, :H], dqkv[:, H:2 * H],
                     dqkv[:, 2 * H:], B, S, nheads, D, 3 * H, 1.0 / math.sqrt(D), rope_cos=cos if fuse_rope else None,
                     rope_sin=sin if fuse_rope else None, transposed=tr, groups=ctx.groups)
        del do
        if not fuse_rope:
            hip.rope_inplace(dqkv, cos, sin, cos.shape[0], nheads, D, 0, H, backward=True)
        if need[4]:
            grads[4] = deliver_wgrad_nt((wo,), hip.transpose(dh1), tr[1] if tr is not None else hip.transpose(o), need[4:5])[0]
        # ---- q | k | v projection
        dxn1 = hip.gemm(dqkv, wT((wq, wk, wv)))                          # [T, H], K = 3H
        if want_qkv_w:
            xn1T = hip.rmsnorm_apply_t(h2, ln1, rstd1) if xn1 is None else hip.transpose(xn1)
            grads[1], grads[2], grads[3] = deliver_wgrad_nt((wq, wk, wv), tr[0] if tr is not None else hip.transpose(dqkv), xn1T,
                                                            need[1:4])
            del xn1T
        del dqkv, tr

        def ln1_run(dw_out, acc):
            holder["dh"] = hip.rmsnorm_bwd(dxn1, h2, ln1, rstd1, dres=dh1, dw_out=dw_out, dw_accumulate=acc)

        if need[0]:
            grads[0] = deliver_vec_grad(ln1, ln1_run)
        else:
            ln1_run(None, False)
        dh = holder["dh"][:Tr].view(B, S, H) if ctx.needs_input_grad[0] else None
        return (dh, None, None, None, None, None, None, None, *grads)


class ReadoutLayerFn(torch.autograd.Function):
    """The LAST LlamaDecoderLayer when only a few rows of its output are read (round 6, opt-in: MLA.readout_rows_only).

    In the diffusion branch the only consumer of the last hidden state is the action read-out (models/vlm/prismatic.py:1115-1126:
    rows k + 2 .. k + 2 + T of every sequence -> FinalLayer); lm_head + CE are computed-but-unused and `output`
    is discarded by the trainer. A decoder layer is row-wise everywhere except in the attention core, so for the last layer:
      all rows : RMSNorm -> q|k|v projection (+RoPE) -> causal attention            (keys / values of every row are attended to)
      read rows: o_proj + residual -> RMSNorm -> gate|up -> SwiGLU -> down + residual   (n rows instead of B * S)
    and in the backward the gradient of the output is non-zero on the read rows only: MLP and o_proj backward run on n rows, the
    attention backward and the q|k|v backward stay dense (d(out) of the attention is a zero matrix with n rows filled in).
    Same mathematics as DecoderLayerFn followed by a row gather -- every loss and every gradient agrees with it to rounding (the small
    GEMMs run on the 128-tile kernel: another accumulation order) -- at 9 of 12 units of the layer's forward GEMM work and 18 of 24 of
    its backward's saved (one unit = rows x H x H): 2.3 % of the step's GEMM FLOPs at 32 layers.
    forward(h [B, S, H], seqlens, cos, sin, nheads, eps, rows int64 [n] (flat indices into B * S), *w) -> [n, H]."""

    @staticmethod
    def forward(ctx, h, seqlens, cos, sin, nheads, eps, rows, *w):
        _check_bf16_cuda(h, *w)
        ln1, wq, wk, wv, wo, ln2, wg, wu, wd = w
        B, S, H = h.shape
        D = H // nheads
        h2 = h.reshape(B * S, H)
        if not h2.is_contiguous():
            h2 = h2.contiguous()
        T, Tp = B * S, _pad_tokens(B * S)
        if Tp != T:
            h2 = torch.cat([h2, h2.new_zeros(Tp - T, H)], 0)
        groups = current_attn_groups()
        # ---- dense half (DecoderLayerFn._fwd up to the attention output)
        xn1, rstd1 = hip.rmsnorm_fwd(h2, ln1, eps)
        qkv = torch.empty((Tp, 3 * H), dtype=BF16, device=h2.device)
        wqkv = cat_view((wq, wk, wv))
        Sr = cos.shape[0]
        if not (wqkv is not None and D == 128 and _ROPE_EPILOGUE and hip.gemm_qkv_rope(xn1, wqkv, qkv, cos, sin, Sr, 2 * H)):
            if wqkv is not None:
                hip.gemm(xn1, wqkv, out=qkv)
            else:
                for i, wi in enumerate((wq, wk, wv)):
                    hip.gemm(xn1, wi, out=qkv[:, i * H:(i + 1) * H])
            hip.rope_inplace(qkv, cos, sin, Sr, nheads, D, 0, H)
        del xn1
        o, lse = hip.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, S, nheads, D, 3 * H, seqlens, 1.0 / math.sqrt(D),
                              rows=Tp, groups=groups)
        # ---- read rows only: zero rows pad n to a multiple of 64 (they stay zero through every row-wise op and add nothing to any
        # weight gradient), so every small GEMM has MFMA-friendly row / reduction counts
        n = rows.numel()
        npad = (n + 63) // 64 * 64
        o_r = hip.gather_rows(o, rows, out_rows=npad)
        h_r = hip.gather_rows(h2, rows, out_rows=npad)
        h1_r = hip.gemm(o_r, wo, residual=h_r)
        xn2_r, rstd2_r = hip.rmsnorm_fwd(h1_r, ln2, eps)
        wgu = cat_view((wg, wu))
        if wgu is not None:
            gu_r = hip.gemm(xn2_r, wgu)
        else:
            I = wg.shape[0]
            gu_r = torch.empty((npad, 2 * I), dtype=BF16, device=h2.device)
            hip.gemm(xn2_r, wg, out=gu_r[:, :I])
            hip.gemm(xn2_r, wu, out=gu_r[:, I:])
        act_r = hip.swiglu_fwd(gu_r)
        out_r = hip.gemm(act_r, wd, residual=h1_r)
        ctx.w, ctx.dims, ctx.aux, ctx.groups, ctx.n = w, (B, S, H, nheads, eps), (seqlens, cos, sin), groups, n
        ctx.save_for_backward(h2, rstd1, qkv, o, lse, rows, o_r, h1_r, rstd2_r, gu_r, act_r)
        return out_r[:n]

    @staticmethod
    def backward(ctx, dout_r):
        ln1, wq, wk, wv, wo, ln2, wg, wu, wd = ctx.w
        B, S, H, nheads, eps = ctx.dims
        seqlens, cos, sin = ctx.aux
        D = H // nheads
        h2, rstd1, qkv, o, lse, rows, o_r, h1_r, rstd2_r, gu_r, act_r = ctx.saved_tensors
        need = ctx.needs_input_grad[7:]
        n, npad, T = ctx.n, o_r.shape[0], h2.shape[0]
        Tr = B * S
        d_r = dout_r.reshape(n, H)
        if npad != n or not d_r.is_contiguous():
            d_r = torch.cat([d_r, d_r.new_zeros(npad - n, H)], 0) if npad != n else d_r.contiguous()
        grads: List[Optional[torch.Tensor]] = [None] * 9
        # ---- MLP on the read rows (dgrad with the weights as they are stored: reduction-major B operand, npad rows make it cheap)
        dact_r = hip.gemm(d_r, wd, b_mode=1)                                     # [npad, I]
        if need[8]:
            grads[8] = deliver_wgrad((wd,), d_r, act_r, need[8:9])[0]
        dgu_r, _ = hip.swiglu_bwd(dact_r, gu_r)
        wgu = cat_view((wg, wu))
        if wgu is not None:
            dxn2_r = hip.gemm(dgu_r, wgu, b_mode=1)                              # [npad, H]
        else:
            I = wg.shape[0]
            dxn2_r = hip.gemm(dgu_r[:, :I].contiguous(), wg, b_mode=1)
            dxn2_r = hip.gemm(dgu_r[:, I:].contiguous(), wu, b_mode=1, residual=dxn2_r)
        if need[6] or need[7]:
            xn2_r, _ = hip.rmsnorm_fwd(h1_r, ln2, eps)
            grads[6], grads[7] = deliver_wgrad((wg, wu), dgu_r, xn2_r, need[6:8])
        holder = {}

        def ln2_run(dw_out, acc):
            holder["dh1"] = hip.rmsnorm_bwd(dxn2_r, h1_r, ln2, rstd2_r, dres=d_r, dw_out=dw_out, dw_accumulate=acc)

        if need[5]:
            grads[5] = deliver_vec_grad(ln2, ln2_run)
        else:
            ln2_run(None, False)
        dh1_r = holder["dh1"]                                                    # [npad, H]: gradient of the residual stream on the read rows
        # ---- attention output projection on the read rows; its input gradient goes back into a dense zero matrix
        do_r = hip.gemm(dh1_r, wo, b_mode=1)
        if need[4]:
            grads[4] = deliver_wgrad((wo,), dh1_r, o_r, need[4:5])[0]
        do = hip.gather_rows(do_r[:n], rows, out_rows=T, scatter=True)
        dres = hip.gather_rows(dh1_r[:n], rows, out_rows=T, scatter=True)        # residual path into the layer input
        # ---- dense half: attention backward, q|k|v backward, input norm backward (DecoderLayerFn.backward)
        dqkv = torch.empty_like(qkv)
        if T != Tr:
            dqkv[Tr:].zero_()
        fuse_rope = cos.shape[0] in (S, B * S) and cos.is_contiguous() and sin.is_contiguous() and cos.dtype == torch.float32
        want_qkv_w = need[1] or need[2] or need[3]
        tr = None
        if _ATTN_BWD_T and fuse_rope and S % 4 == 0 and want_qkv_w:
            tr = (torch.empty((3 * H, T), dtype=BF16, device=qkv.device), torch.empty((H, T), dtype=BF16, device=qkv.device))
            if T != Tr:
                tr[0][:, Tr:].zero_()
                tr[1][:, Tr:].zero_()
        hip.attn_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], o, do, lse, seqlens, dqkv[:, :H], dqkv[:, H:2 * H],
                     dqkv[:, 2 * H:], B, S, nheads, D, 3 * H, 1.0 / math.sqrt(D), rope_cos=cos if fuse_rope else None,
                     rope_sin=sin if fuse_rope else None, transposed=tr, groups=ctx.groups)
        del do
        if not fuse_rope:
            hip.rope_inplace(dqkv, cos, sin, cos.shape[0], nheads, D, 0, H, backward=True)
        wqkv = cat_view((wq, wk, wv))
        dxn1 = hip.gemm(dqkv, hip.transpose(wqkv if wqkv is not None else torch.cat([wq, wk, wv], 0)))
        if want_qkv_w:
            xn1T = hip.rmsnorm_apply_t(h2, ln1, rstd1)
            grads[1], grads[2], grads[3] = deliver_wgrad_nt((wq, wk, wv), tr[0] if tr is not None else hip.transpose(dqkv), xn1T, need[1:4])
            del xn1T
        del dqkv, tr

        def ln1_run(dw_out, acc):
            holder["dh"] = hip.rmsnorm_bwd(dxn1, h2, ln1, rstd1, dres=dres, dw_out=dw_out, dw_accumulate=acc)

        if need[0]:
            grads[0] = deliver_vec_grad(ln1, ln1_run)
        else:
            ln1_run(None, False)
        dh = holder["dh"][:Tr].view(B, S, H) if ctx.needs_input_grad[0] else None
        return (dh, None, None, None, None, None, None, *grads)


def decoder_layer_readout(h, seqlens, cos, sin, nheads, eps, rows, weights):
    """Rows `rows` (flat indices into B * S) of LlamaDecoderLayer(h): see ReadoutLayerFn."""
    return ReadoutLayerFn.apply(h, seqlens, cos, sin, nheads, eps, rows, *weights)


def decoder_layer(h, seqlens, cos, sin, nheads, eps, save_level, weights, fold_io=None):
    return DecoderLayerFn.apply(h, seqlens, cos, sin, nheads, eps, save_level, fold_io, *weights)


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


class GatherRowsSumFn(torch.autograd.Function):
    """out[r] = src[idx[r]] for r < len(idx), zero rows up to `out_rows`; idx MAY REPEAT (several point centres projecting into one
    image patch, models/mla/fuser/contrastive.py:185-207). Backward: d src[s] = sum of dy[r] over idx[r] == s, accumulated in fp32
    in ascending r by the deterministic embedding-backward kernel and rounded once -- torch's index_select backward adds bf16 values
    with atomics in arrival order, which made the step's gradient norm reproducible only to 1e-5."""

    @staticmethod
    def forward(ctx, src2d, idx, out_rows):
        _check_bf16_cuda(src2d)
        ctx.save_for_backward(idx)
        ctx.n_src = src2d.shape[0]
        return hip.gather_rows(src2d.contiguous(), idx, out_rows=out_rows)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n = idx.numel()
        g32 = torch.zeros((ctx.n_src, dy.shape[1]), dtype=torch.float32, device=dy.device)
        hip.embedding_bwd(idx, dy[:n].contiguous(), g32)
        return hip.cast_f32_to_bf16(g32), None, None


def gather_rows_sum(src2d, idx, out_rows=None):
    idx = idx.contiguous()
    return GatherRowsSumFn.apply(src2d, idx, idx.numel() if out_rows is None else out_rows)


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


# ------------------------------------------------------------------------------------------------- generation heads
_seed_counter = [0]


def next_seed() -> int:
    """Host-side counter-based seed for the dropout kernels (no device sync). Derived from torch's seed so that
    torch.manual_seed() makes a run reproducible; the kernels hash (seed, element index)."""
    _seed_counter[0] += 1
    x = (torch.initial_seed() + 0x9E3779B97F4A7C15 * _seed_counter[0]) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 27
    return x


def param_view(param: torch.Tensor, lo: Optional[int] = None, hi: Optional[int] = None, shape=None) -> torch.Tensor:
    """A row-slice / reshape of a parameter that still delivers its weight gradient into the matching region of
    ``param.main_grad`` (nn.MultiheadAttention's packed in_proj_weight, Conv1d(k=1) weights used as matrices)."""
    v = param if lo is None else param[lo:hi]
    if shape is not None:
        v = v.view(shape)
    mg = getattr(param, "main_grad", None)
    if mg is not None:
        g = mg if lo is None else mg[lo:hi]
        v.main_grad = g.view(shape) if shape is not None else g
        if not hasattr(param, "_mg_regions"):
            param._mg_regions = {}    # region -> "main_grad region holds a contribution from this accumulation window"; cleared by
                                      # FlatUnit.begin_step (parameters without a FlatUnit: ops.reset_main_grad_state before every
                                      # window), set only INSIDE the backward that writes the region (_mark_touched), so an
                                      # eval / no_grad forward or an unused branch leaves the parameter untouched
        v._mg_region = (param, (lo, hi))
    return v


class DropoutAddFn(torch.autograd.Function):
    """y = residual + dropout(x, p) (residual optional); the mask is regenerated from (seed, index) in backward."""

    @staticmethod
    def forward(ctx, x, residual, p, seed):
        _check_bf16_cuda(x, residual)
        ctx.p, ctx.seed, ctx.has_res = p, seed, residual is not None
        return hip.dropout_fwd(x.contiguous(), residual.contiguous() if residual is not None else None, p, seed)

    @staticmethod
    def backward(ctx, dy):
        dyc = dy.contiguous()
        dx = hip.dropout_bwd(dyc, ctx.p, ctx.seed) if ctx.p > 0 else dyc
        return dx, (dyc if ctx.has_res else None), None, None


def dropout_add(x, residual, p, training):
    p = p if training else 0.0
    if p == 0.0 and residual is None:
        return x
    return DropoutAddFn.apply(x, residual, p, next_seed() if p > 0 else 0)


class ScaleBatchFn(torch.autograd.Function):
    """timm DropPath: x[b] * scale[b] with scale = bernoulli(keep) / keep (models/mla/generation/models.py:45,63-64)."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.save_for_backward(scale)
        return hip.scale_batch(x.contiguous(), scale)

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        return hip.scale_batch(dy.contiguous(), scale), None


def drop_path(x, p, training):
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    scale = torch.empty(x.shape[0], dtype=torch.float32, device=x.device).bernoulli_(keep).div_(keep)
    return ScaleBatchFn.apply(x, scale)


class MHACoreFn(torch.autograd.Function):
    """softmax(Q K^T / sqrt(hd)) V for every (sample, head) of nn.MultiheadAttention (batch_first), attention dropout
    included -- torch/nn/functional.py multi_head_attention_forward as called by nn.TransformerDecoderLayer
    (models/mla/generation/models.py:103-122) and TransformerBlock (:44,62).

    qsrc: [B, Sq, 3E] packed q|k|v (self-attention, kvsrc None) or [B, Sq, E] (cross-attention, kvsrc [B, Skp, 2E] packed
    k|v). Only the first ``nvalid`` key rows take part (the rest is alignment padding). The QK^T / PV products and their four
    gradients are batched MFMA GEMMs over (B, heads); scores are fp32."""

    @staticmethod
    def forward(ctx, qsrc, kvsrc, nheads, nvalid, p, seed):
        _check_bf16_cuda(qsrc, kvsrc)
        qsrc = qsrc.contiguous()
        B, Sq = qsrc.shape[0], qsrc.shape[1]
        if kvsrc is None:
            E = qsrc.shape[2] // 3
            q, k, v = qsrc[..., :E], qsrc[..., E:2 * E], qsrc[..., 2 * E:]
            Sk = Sq
        else:
            kvsrc = kvsrc.contiguous()
            E = qsrc.shape[2]
            q, k, v = qsrc, kvsrc[..., :E], kvsrc[..., E:]
            Sk = kvsrc.shape[1]
        hd = E // nheads
        scale = 1.0 / math.sqrt(hd)
        dev = qsrc.device
        scores = torch.empty((B, nheads, Sq, Sk), dtype=torch.float32, device=dev)
        sS = (nheads * Sq * Sk, Sq * Sk)
        hip.gemm_batched(q, k, scores, M=Sq, N=Sk, K=hd, lda=q.stride(1), ldb=k.stride(1), ldc=Sk, alpha=scale, n_outer=B,
                         n_inner=nheads, sA=(q.stride(0), hd), sB=(k.stride(0), hd), sC=sS)
        P, Pd = hip.softmax_rows_fwd(scores, nvalid, p, seed)
        del scores
        o = torch.empty((B, Sq, E), dtype=BF16, device=dev)
        hip.gemm_batched(Pd, v, o, M=Sq, N=hd, K=Sk, lda=Sk, ldb=v.stride(1), ldc=E, b_mode=1, n_outer=B, n_inner=nheads, sA=sS,
                         sB=(v.stride(0), hd), sC=(Sq * E, hd))
        ctx.save_for_backward(qsrc, kvsrc, P)
        ctx.meta = (nheads, nvalid, p, seed, E, Sk)
        return o

    @staticmethod
    def backward(ctx, do):
        qsrc, kvsrc, P = ctx.saved_tensors
        nheads, nvalid, p, seed, E, Sk = ctx.meta
        do = do.contiguous()
        B, Sq = qsrc.shape[0], qsrc.shape[1]
        hd = E // nheads
        scale = 1.0 / math.sqrt(hd)
        dqsrc = torch.empty_like(qsrc)
        if kvsrc is None:
            q, k, v = qsrc[..., :E], qsrc[..., E:2 * E], qsrc[..., 2 * E:]
            dq, dk, dv = dqsrc[..., :E], dqsrc[..., E:2 * E], dqsrc[..., 2 * E:]
            dkvsrc = None
        else:
            dkvsrc = torch.empty_like(kvsrc)
            q, k, v = qsrc, kvsrc[..., :E], kvsrc[..., E:]
            dq, dk, dv = dqsrc, dkvsrc[..., :E], dkvsrc[..., E:]
        sS = (nheads * Sq * Sk, Sq * Sk)
        # dPd = dO V^T, kept in fp32 (transient): the softmax backward subtracts the row's weighted mean from it, so a bf16 dP would
        # keep only ~8 bits of what survives the subtraction (csrc/gen.hip softmax_rows_bwd_kernel; torch's fused attention, which the
        # reference runs under autocast, does not round dP either)
        dPd = torch.empty(P.shape, dtype=torch.float32, device=P.device)
        hip.gemm_batched(do, v, dPd, M=Sq, N=Sk, K=hd, lda=E, ldb=v.stride(1), ldc=Sk, n_outer=B, n_inner=nheads, sA=(Sq * E, hd),
                         sB=(v.stride(0), hd), sC=sS)
        # dV = Pd^T dO  (Pd regenerated from P and the dropout hash: Pd = P * mask / (1 - p))
        if p > 0:
            Pd = hip.dropout_fwd(P, None, p, seed)
        else:
            Pd = P
        hip.gemm_batched(Pd, do, dv, M=Sk, N=hd, K=Sq, lda=Sk, ldb=E, ldc=dv.stride(1), a_mode=1, b_mode=1, n_outer=B, n_inner=nheads,
                         sA=sS, sB=(Sq * E, hd), sC=(dv.stride(0), hd))
        del Pd
        dS = hip.softmax_rows_bwd(dPd, P, nvalid, p, seed)
        del dPd
        # dQ = scale * dS K ; dK = scale * dS^T Q
        hip.gemm_batched(dS, k, dq, M=Sq, N=hd, K=Sk, lda=Sk, ldb=k.stride(1), ldc=dq.stride(1), b_mode=1, alpha=scale, n_outer=B,
                         n_inner=nheads, sA=sS, sB=(k.stride(0), hd), sC=(dq.stride(0), hd))
        hip.gemm_batched(dS, q, dk, M=Sk, N=hd, K=Sq, lda=Sk, ldb=q.stride(1), ldc=dk.stride(1), a_mode=1, b_mode=1, alpha=scale,
                         n_outer=B, n_inner=nheads, sA=sS, sB=(q.stride(0), hd), sC=(dk.stride(0), hd))
        return dqsrc, dkvsrc, None, None, None, None


def mha_core(qsrc, kvsrc, nheads, nvalid, p, training):
    p = p if training else 0.0
    return MHACoreFn.apply(qsrc, kvsrc, nheads, nvalid, p, next_seed() if p > 0 else 0)


class SeqMeanFn(torch.autograd.Function):
    """x.mean(dim=1) over [B, S, C] (PointCloudGenerationModule.forward models/mla/generation/models.py:359)."""

    @staticmethod
    def forward(ctx, x):
        _check_bf16_cuda(x)
        ctx.S = x.shape[1]
        return hip.seqmean_fwd(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return hip.seqmean_bwd(dy.contiguous(), ctx.S)


class BatchNormTrainFn(torch.autograd.Function):
    """nn.BatchNorm1d in training mode over rows [rows, C] (+ ReLU): batch statistics forward and the full backward
    (future_predictor, models/mla/generation/models.py:333-338). Returns (y, mean, biased var) -- the module updates its
    running statistics from them."""

    @staticmethod
    def forward(ctx, x2, weight, bias, eps, relu):
        _check_bf16_cuda(x2, weight, bias)
        mean, var = hip.colstats(x2)
        y = hip.bn_apply(x2, mean, var, weight, bias, eps, relu=relu)
        ctx.save_for_backward(x2, mean, var, y)
        ctx.weight, ctx.bias, ctx.eps, ctx.relu = weight, bias, eps, relu
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x2, mean, var, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            dy = hip.act_bwd(dy, y, hip.ACT_RELU)      # relu'(pre) == relu'(post) as a 0/1 gate
        C = x2.shape[1]
        dwb = torch.empty((2, C), dtype=torch.float32, device=x2.device)
        dx = hip.bn_bwd(dy, x2, mean, var, ctx.weight, ctx.eps, dw=dwb[0], db=dwb[1])
        dw = _deliver_small(ctx.weight, dwb[0]) if ctx.needs_input_grad[1] else None
        db = _deliver_small(ctx.bias, dwb[1]) if ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None


class ChamferFn(torch.autograd.Function):
    """chamfer_distance_l2 (models/mla/generation/gen_loss.py:12-18): Euclidean nearest-neighbour distances both ways."""

    @staticmethod
    def forward(ctx, pred, gt):
        pred32 = pred.to(torch.float32).contiguous()
        gt32 = gt.to(torch.float32).contiguous()
        loss, d1, i1, d2, i2 = hip.chamfer_fwd(pred32, gt32)
        ctx.save_for_backward(pred32, gt32, d1, i1, d2, i2)
        ctx.in_dtype = pred.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        pred32, gt32, d1, i1, d2, i2 = ctx.saved_tensors
        d = hip.chamfer_bwd(pred32, gt32, d1, i1, d2, i2, g.to(torch.float32).contiguous())
        return d.to(ctx.in_dtype), None


class ImageGenLossFn(torch.autograd.Function):
    """Image generation loss for the all-true ROI (use_roi False): generated = 0.05 * current + clip * tanh(delta_raw)
    (ImageGenerationModule._generate_generated_patches models.py:264-283 with mask == 1), loss = MSE + 0.5 L1 vs the next
    frame's patches - 0.1 mean|delta| (models/vlm/prismatic.py:780-816). Returns (loss, mse, l1, mean|delta|)."""

    @staticmethod
    def forward(ctx, delta_raw, curr_img, next_img, ps, clip):
        _check_bf16_cuda(delta_raw)
        delta_raw = delta_raw.contiguous()
        sums = hip.imgloss_fwd(delta_raw, curr_img, next_img, ps, clip)
        n = float(delta_raw.numel() // delta_raw.shape[-1] * 3 * ps * ps)     # valid columns only (rows may be padded)
        parts = sums / n
        loss = parts[0] + 0.5 * parts[1] - 0.1 * parts[2]
        ctx.save_for_backward(delta_raw, curr_img, next_img)
        ctx.ps, ctx.clip = ps, clip
        ctx.mark_non_differentiable(parts)
        return loss, parts

    @staticmethod
    def backward(ctx, g, _gp):
        delta_raw, curr_img, next_img = ctx.saved_tensors
        d = hip.imgloss_bwd(delta_raw, curr_img, next_img, ctx.ps, ctx.clip, g.to(torch.float32).reshape(1).contiguous())
        return d, None, None, None, None


class PaddedLinearFn(torch.autograd.Function):
    """y = x W^T + b with the output width padded up to a multiple of 64 (pad columns are exactly zero). For heads whose width
    breaks the 16-byte row alignment the MFMA GEMMs need on the backward operands -- mae_delta_head has 3*42*42 = 5292 outputs
    (models/mla/generation/models.py:125-127): padded to 5312, forward, dgrad and wgrad all stay on the k-contiguous MFMA kernel
    instead of the SIMT fallback. The parameter keeps its reference shape; a zero-padded bf16 copy is rebuilt per step."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _check_bf16_cuda(x, weight, bias)
        x2 = _as2d(x)
        N, K = weight.shape
        Np = (N + 63) // 64 * 64
        wp = torch.zeros((Np, K), dtype=BF16, device=x.device)
        wp[:N].copy_(weight)
        bp = None
        if bias is not None:
            bp = torch.zeros(Np, dtype=BF16, device=x.device)
            bp[:N].copy_(bias)
        out = hip.gemm(x2, wp, bias=bp)
        ctx.save_for_backward(x2, wp)
        ctx.weight, ctx.bias, ctx.in_shape = weight, bias, x.shape
        return out.view(*x.shape[:-1], Np)

    @staticmethod
    def backward(ctx, dy):
        x2, wp = ctx.saved_tensors
        N = ctx.weight.shape[0]
        dy2 = _as2d(dy)
        ni = ctx.needs_input_grad
        dx = hip.gemm(dy2, hip.transpose(wp)).view(ctx.in_shape) if ni[0] else None
        dw = None
        if ni[1]:
            dw = deliver_wgrad_nt((ctx.weight,), hip.transpose(dy2)[:N], hip.transpose(x2), (True,))[0]
        db = None
        if ctx.bias is not None and ni[2]:
            db = deliver_vec_grad(ctx.bias, lambda out, acc: hip.colsum(dy2[:, :N], out, acc))
        return dx, dw, db


# ------------------------------------------------------------------------------------------------- vision tokenizer (stage "pretrain")
class AvgPoolTokensFn(torch.autograd.Function):
    """F.avg_pool2d(k = s = cs) over channel-last token rows [B*gh*gw, C] (models/mla/image/vision_tokenizer.py:28)."""

    @staticmethod
    def forward(ctx, x, B, gh, gw, cs):
        _check_bf16_cuda(x)
        ctx.dims = (B, gh, gw, cs)
        return hip.avgpool_tokens(x.contiguous(), B, gh, gw, cs)

    @staticmethod
    def backward(ctx, dy):
        B, gh, gw, cs = ctx.dims
        return hip.avgpool_tokens_bwd(dy.contiguous(), None, B, gh, gw, cs), None, None, None, None


class LocalAttnFn(torch.autograd.Function):
    """3x3-window attention of LocalAttention.forward (vision_tokenizer.py:26-47): q [windows, C], kv [tokens, 2C] -> [windows, C]."""

    @staticmethod
    def forward(ctx, q, kv, B, gh, gw, cs, heads, scale):
        _check_bf16_cuda(q, kv)
        q, kv = q.contiguous(), kv.contiguous()
        ctx.save_for_backward(q, kv)
        ctx.dims = (B, gh, gw, cs, heads, scale)
        return hip.local_attn(q, kv, B, gh, gw, cs, heads, scale)

    @staticmethod
    def backward(ctx, dout):
        q, kv = ctx.saved_tensors
        dq, dkv = hip.local_attn_bwd(q, kv, dout.contiguous(), *ctx.dims)
        return dq, dkv, None, None, None, None, None, None


class BmmNTFn(torch.autograd.Function):
    """C[b] = alpha * A[b] @ B[b]^T for bf16 [nb, M, K] x [nb, N, K] -> fp32 [nb, M, N] (torch.bmm(a, b.transpose(1, 2)) / T of
    TactileContrastiveLoss, models/mla/fuser/contrastive.py:248,253). Tiny M (one row per arm): the batched GEMM falls back to
    its SIMT path where the MFMA alignment rules do not hold."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        _check_bf16_cuda(a, b)
        a, b = a.contiguous(), b.contiguous()
        nb, M, K = a.shape
        N = b.shape[1]
        out = torch.empty((nb, M, N), dtype=torch.float32, device=a.device)
        hip.gemm_batched(a, b, out, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, alpha=alpha, n_outer=nb, n_inner=1, sA=(M * K, 0), sB=(N * K, 0),
                         sC=(M * N, 0))
        ctx.save_for_backward(a, b)
        ctx.alpha = alpha
        return out

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        nb, M, K = a.shape
        N = b.shape[1]
        dcb = dc.to(BF16).contiguous()
        da = torch.empty_like(a)
        db = torch.empty_like(b)
        # dA = alpha * dC B   (B stored [N(k)][K(n)] -> reduction-major)     dB = alpha * dC^T A  (both reduction-major)
        hip.gemm_batched(dcb, b, da, M=M, N=K, K=N, lda=N, ldb=K, ldc=K, b_mode=1, alpha=ctx.alpha, n_outer=nb, n_inner=1, sA=(M * N, 0),
                         sB=(N * K, 0), sC=(M * K, 0))
        hip.gemm_batched(dcb, a, db, M=N, N=K, K=M, lda=N, ldb=K, ldc=K, a_mode=1, b_mode=1, alpha=ctx.alpha, n_outer=nb, n_inner=1,
                         sA=(M * N, 0), sB=(M * K, 0), sC=(N * K, 0))
        return da, db, None


class ImageGenRoiLossFn(torch.autograd.Function):
    """Image generation with a partial ROI (use_roi=True): ROI patches follow `0.05 * current + delta`, the others are the
    alpha-blend of the current patch with its translated (warped) copy + delta (models/mla/generation/models.py:226-286); loss =
    MSE + 0.5 L1 on ROI elements + 0.01 L1 on background elements - 0.1 mean|delta| (models/vlm/prismatic.py:786-816; a term whose
    set is empty is dropped). Returns (loss, parts = [roi mse, roi l1, bg l1, mean|delta|])."""

    @staticmethod
    def forward(ctx, delta_raw, a_raw, o_raw, roi_mask, curr_img, next_img, ps, clip, shift):
        _check_bf16_cuda(delta_raw, a_raw, o_raw)
        delta_raw, a_raw, o_raw = delta_raw.contiguous(), a_raw.contiguous(), o_raw.contiguous()
        B, npatch = delta_raw.shape[0], delta_raw.shape[1]
        pd = 3 * ps * ps
        roi = roi_mask.reshape(B * npatch).to(torch.uint8).contiguous()
        a2, o2 = a_raw.reshape(B * npatch, -1), o_raw.reshape(B * npatch, -1)
        sums = hip.imgroi_fwd(delta_raw, a2, o2, roi, curr_img, next_img, ps, clip, shift)
        n_roi = roi.sum().to(torch.float32) * pd
        n_all = float(B * npatch * pd)
        n_bg = n_all - n_roi
        inv_roi = torch.where(n_roi > 0, 1.0 / n_roi.clamp(min=1.0), torch.zeros_like(n_roi))
        inv_bg = torch.where(n_bg > 0, 1.0 / n_bg.clamp(min=1.0), torch.zeros_like(n_bg))
        parts = torch.stack([sums[0] * inv_roi, sums[1] * inv_roi, sums[2] * inv_bg, sums[3] / n_all])
        loss = parts[0] + 0.5 * parts[1] + 0.01 * parts[2] - 0.1 * parts[3]
        ctx.save_for_backward(delta_raw, a2, o2, roi, curr_img, next_img, torch.stack([inv_roi, 0.01 * inv_bg, inv_roi.new_tensor(-0.1 / n_all)]))
        ctx.meta = (ps, clip, shift, a_raw.shape, o_raw.shape)
        ctx.mark_non_differentiable(parts)
        return loss, parts

    @staticmethod
    def backward(ctx, g, _gp):
        delta_raw, a2, o2, roi, curr_img, next_img, k = ctx.saved_tensors
        ps, clip, shift, ashape, oshape = ctx.meta
        coef = (k * g.to(torch.float32)).contiguous()
        dd, da, do = hip.imgroi_bwd(delta_raw, a2, o2, roi, curr_img, next_img, ps, clip, shift, coef)
        da_full = torch.zeros(a2.shape, dtype=BF16, device=a2.device)
        da_full[:, 0] = da.to(BF16)
        do_full = torch.zeros(o2.shape, dtype=BF16, device=o2.device)
        do_full[:, :2] = do.to(BF16)
        return dd, da_full.view(ashape), do_full.view(oshape), None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------- point tokenizer (stage "pretrain")
class LgaPrepFn(torch.autograd.Function):
    """Neighbour / centre feature gather + positional embedding of LGA.forward (Point_PN.py:115-158): differentiable in the point
    features only (coordinates and the sin/cos embedding are constants)."""

    @staticmethod
    def forward(ctx, xyz, feats, fps_idx, knn_idx, alpha, beta):
        feats = feats.contiguous()
        rows, lc_xyz = hip.lga_prep(xyz, feats, fps_idx, knn_idx, alpha, beta)
        ctx.save_for_backward(fps_idx, knn_idx)
        ctx.shape = feats.shape
        ctx.mark_non_differentiable(lc_xyz)
        return rows, lc_xyz

    @staticmethod
    def backward(ctx, drows, _dl):
        fps_idx, knn_idx = ctx.saved_tensors
        B, N, C = ctx.shape
        d32 = hip.lga_prep_bwd(drows.contiguous(), fps_idx, knn_idx, B, N, C)
        return None, hip.cast_f32_to_bf16(d32), None, None, None, None


class MaxPoolKFn(torch.autograd.Function):
    """max over the K neighbours of every group (Pooling.forward Point_PN.py:166-169)."""

    @staticmethod
    def forward(ctx, rows, groups, K):
        rows = rows.contiguous()
        ctx.save_for_backward(rows)
        ctx.dims = (groups, K)
        return hip.maxpool_k(rows, groups, K)

    @staticmethod
    def backward(ctx, dy):
        (rows,) = ctx.saved_tensors
        return hip.maxpool_k_bwd(rows, dy.contiguous(), *ctx.dims), None, None
