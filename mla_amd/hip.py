"""ctypes binding of libmla_hip.so (C-ABI declared in include/mla_hip.h).

Every wrapper launches on torch's current HIP stream, never synchronises, and raises RuntimeError on a
non-zero return code. There is deliberately NO fallback: if the shared library is missing or a kernel rejects its
arguments the product path fails loudly (oracle/ is test infrastructure only).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_size_t, c_void_p
from typing import Optional

import torch

_LIB_PATH = os.environ.get("MLA_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmla_hip.so")   # override: A/B experiments
_lib = None

ACT_GELU_ERF, ACT_GELU_TANH, ACT_RELU, ACT_SILU = 0, 1, 2, 3

# split-K tail of the 256x256 GEMM (mla_gemm_bf16_ws): on by default; MLA_GEMM_SPLITK=0 turns it off for A/B measurements
SPLITK = os.environ.get("MLA_GEMM_SPLITK", "1") != "0"
SPLITK_WS_BYTES = int(os.environ.get("MLA_GEMM_SPLITK_WS_MB", "64")) << 20     # fp32 partial tiles of the K-slices (256 KiB each); 256 MiB would also split the 688-tile down-projection wgrad (176 tail tiles x 4): measured 1.2 % SLOWER per launch

# bench.py sets this to a list to time every GEMM launch with HIP events on the launch stream (roofline.achieved)
GEMM_PROFILE = None

# name -> argtypes (restype is always int unless noted) -- mirrors include/mla_hip.h
_SIGNATURES = {
    "mla_query": [c_int],
    "mla_selftest": [c_void_p, c_void_p, c_void_p, c_void_p],
    "mla_dispatch_probe": [c_void_p, c_int, c_int, c_void_p],
    "mla_calib_mfma": [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p],
    "mla_gemv_bf16": [c_void_p, c_longlong, c_void_p, c_longlong, c_void_p, c_longlong, c_longlong, c_int, c_void_p, c_longlong, c_int, c_int,
                      c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p],
    "mla_attn_decode": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_longlong, c_float,
                        c_void_p],
    "mla_gemm_bf16": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                      c_int, c_int, c_int, c_int, c_float, c_int, c_void_p],
    "mla_gemm_bf16_ws": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                         c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_size_t, c_void_p],
    "mla_gemm_qkv_rope": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                          c_void_p],
    "mla_gemm_bf16_ws_sq": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_size_t,
                            c_void_p, c_int, c_void_p, c_void_p],
    "mla_sum_partials": [c_void_p, c_int, c_void_p, c_int, c_void_p],
    "mla_gemm_sq_slots": [c_int, c_int, c_int, c_size_t],
    "mla_gemm_kloop": [c_int],
    "mla_gemm_cus": [c_int],
    "mla_side_traffic": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_void_p],
    "mla_gemm_gateup_swiglu": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_longlong,
                               c_void_p],
    "mla_gemm_dact_swiglu_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_longlong,
                                 c_void_p],
    "mla_rmsnorm_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "mla_rmsnorm_prep": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "mla_gemm_res_norm": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                          c_int, c_void_p, c_size_t, c_void_p],
    "mla_gemm_qkv_rope_rs": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                             c_void_p, c_int, c_float, c_void_p, c_void_p],
    "mla_gemm_gateup_swiglu_rs": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_longlong,
                                  c_void_p, c_int, c_float, c_void_p, c_void_p],
    "mla_rmsnorm_bwd_blocks": [c_int],
    "mla_timm_rmsnorm_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "mla_timm_rmsnorm_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                             c_size_t, c_void_p],
    "mla_rmsnorm_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                        c_size_t, c_void_p],
    "mla_colsum_blocks": [c_int],
    "mla_colsum_bf16": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p],
    "mla_layernorm_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "mla_rope_inplace": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "mla_swiglu_fwd": [c_void_p, c_void_p, c_longlong, c_int, c_void_p],
    "mla_swiglu_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_void_p],
    "mla_act_fwd": [c_void_p, c_void_p, c_longlong, c_int, c_void_p],
    "mla_act_bwd": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_void_p],
    "mla_cast_f32_to_bf16": [c_void_p, c_void_p, c_longlong, c_void_p],
    "mla_cast_bf16_to_f32": [c_void_p, c_void_p, c_longlong, c_void_p],
    "mla_add_bf16": [c_void_p, c_void_p, c_void_p, c_longlong, c_void_p],
    "mla_embedding_fwd": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p],
    "mla_embedding_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p],
    "mla_adamw_step_groups": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_longlong, c_float, c_float, c_float, c_float,
                              c_float, c_int, c_void_p, c_void_p],
    "mla_adamw_step": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_float, c_float, c_float, c_float,
                       c_float, c_int, c_void_p, c_void_p],
    "mla_sumsq_f32": [c_void_p, c_longlong, c_void_p, c_int, c_void_p, c_size_t, c_void_p],
    "mla_clip_coef": [c_void_p, c_float, c_void_p, c_void_p, c_void_p],
    "mla_q_sample": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mla_attn_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_longlong,
                     c_longlong, c_float, c_void_p],
    "mla_attn_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                     c_void_p, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_float, c_void_p, c_void_p, c_void_p, c_longlong,
                     c_void_p],
    "mla_attn_bwd_t": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                       c_void_p, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_float, c_void_p, c_void_p,
                       c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p],
    "mla_attn_bwd_sync_ints": [c_int, c_int],                # returns long long (restype fixed up in lib())
    "mla_attn_fwd_g": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_longlong,
                       c_longlong, c_float, c_int, c_int, c_void_p, c_void_p],
    "mla_attn_bwd_g": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                       c_void_p, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_float, c_void_p, c_void_p,
                       c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_int, c_int, c_void_p, c_int, c_void_p],
    "mla_attn_bwd_ws_bytes": [c_int, c_int, c_int],          # returns long long (restype fixed up in lib())
    "mla_attn_bwd_ws": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                        c_void_p, c_int, c_int, c_int, c_int, c_longlong, c_longlong, c_float, c_void_p, c_void_p,
                        c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_longlong, c_void_p],
    "mla_ce_fwd": [c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_void_p, c_int, c_int, c_longlong, c_void_p],
    "mla_ce_bwd": [c_void_p, c_int, c_longlong, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_longlong, c_int, c_int,
                   c_longlong, c_void_p],
    "mla_infonce_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    "mla_l2norm_fwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p],
    "mla_l2norm_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
}


def exported_symbols():
    """Names every build of libmla_hip.so must export (checked by the CPU test-suite)."""
    return sorted(list(_SIGNATURES.keys()) + ["mla_last_error", "mla_gemm_source_id"] + list(_EXTRA_SIGNATURES.keys()))


_EXTRA_SIGNATURES = {}  # filled by optional kernel groups (pointcloud / vision) below


def register_signatures(sigs):
    _EXTRA_SIGNATURES.update(sigs)
    global _lib
    if _lib is not None:
        for name, argt in sigs.items():
            fn = getattr(_lib, name)
            fn.argtypes = argt
            fn.restype = c_int


def lib():
    """Load the shared library (once). Raises if it is absent -- there is no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(mla_amd has no CPU fallback)")
        L = ctypes.CDLL(_LIB_PATH)
        L.mla_last_error.restype = ctypes.c_char_p
        L.mla_last_error.argtypes = []
        L.mla_gemm_source_id.restype = ctypes.c_char_p
        L.mla_gemm_source_id.argtypes = []
        for name, argt in {**_SIGNATURES, **_EXTRA_SIGNATURES}.items():
            fn = getattr(L, name)
            fn.argtypes = argt
            fn.restype = c_int
        L.mla_attn_bwd_ws_bytes.restype = c_longlong
        L.mla_attn_bwd_sync_ints.restype = c_longlong
        _lib = L
    return _lib


def gemm_cus(n: int = -1) -> int:
    """CUs the GEMM launches plan their split-K tails for (mla_gemm_cus): n >= 8 (multiple of 8) sets, 0 = device count, < 0 queries."""
    return lib().mla_gemm_cus(n)


def side_traffic(a32, b32, out32, blocks: int, lds_bytes: int = 0, sleep_ticks: int = 0):
    """Stand-in for a collective's kernel on the CURRENT stream (mla_side_traffic): out = a + b streamed by `blocks` workgroups."""
    call("mla_side_traffic", _p(a32), _p(b32), _p(out32), a32.numel(), blocks, lds_bytes, sleep_ticks)


def gemm_source_id() -> str:
    """sha256[:16] over the gemm256 kernel family's sources the loaded library was built from (build.sh stamps it)."""
    return lib().mla_gemm_source_id().decode()


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def _check(rc: int, name: str):
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib().mla_last_error().decode()}")


def _req(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor must live on the GPU (mla_amd has no CPU path)")


def call(name: str, *args):
    """Raw call helper: appends the current stream and checks the return code."""
    rc = getattr(lib(), name)(*args, _stream())
    _check(rc, name)


_ws_cache = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Per-(device, stream) scratch buffer owned by torch (kernels never allocate)."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 24), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# --------------------------------------------------------------------------------------------- GEMM
def gemm(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, *, a_mode: int = 0, b_mode: int = 0,
         M: Optional[int] = None, N: Optional[int] = None, K: Optional[int] = None, lda: Optional[int] = None,
         ldb: Optional[int] = None, ldc: Optional[int] = None, bias: Optional[torch.Tensor] = None,
         residual: Optional[torch.Tensor] = None, ldr: Optional[int] = None, out_dtype=torch.bfloat16,
         accumulate: bool = False, alpha: float = 1.0, force_generic: int = 0) -> torch.Tensor:
    """C[M,N] = alpha * sum_k Aop(m,k) Bop(n,k) (+bias[n]) (+residual[m,n]) (+C if accumulate).

    a_mode/b_mode 0: operand stored [rows, K] (k contiguous); 1: stored [K, rows] (reduction-major).
    force_generic: 0 = auto (256x256 kernel for large k-contiguous shapes, else 128x128, else SIMT fallback),
    1 = SIMT fallback, 2 = never use the 256x256 kernel, 3 = force the 256x256 kernel for any mode (tests / A-B).
    2-D tensors with unit inner stride; leading dimension taken from stride(0).
    """
    _req(a, torch.bfloat16, "gemm a")
    _req(b, torch.bfloat16, "gemm b")
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    if M is None:
        M = a.shape[0] if a_mode == 0 else a.shape[1]
    if K is None:
        K = a.shape[1] if a_mode == 0 else a.shape[0]
    if N is None:
        N = b.shape[0] if b_mode == 0 else b.shape[1]
    kb = b.shape[1] if b_mode == 0 else b.shape[0]
    if kb != K:
        raise ValueError(f"gemm: reduction mismatch {K} vs {kb}")
    lda = a.stride(0) if lda is None else lda
    ldb = b.stride(0) if ldb is None else ldb
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.stride(-1) == 1
    out_fp32 = 1 if out.dtype == torch.float32 else 0
    if not out_fp32:
        _req(out, torch.bfloat16, "gemm out")
    ldc = out.stride(0) if ldc is None else ldc
    if residual is not None:
        _req(residual, torch.bfloat16, "gemm residual")
        ldr = residual.stride(0) if ldr is None else ldr
    if bias is not None:
        _req(bias, torch.bfloat16, "gemm bias")
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()   # torch's current stream == the stream the kernel is launched on (see _stream())
    if SPLITK and (force_generic & 15) == 0 and M >= 256 and N >= 256:
        ws = workspace(SPLITK_WS_BYTES, a.device)      # per (device, stream) scratch owned by torch
        call("mla_gemm_bf16_ws", _p(a), _p(b), _p(out), _p(residual), _p(bias), M, N, K, lda, ldb, ldc, ldr or 0, a_mode, b_mode,
             out_fp32, 1 if accumulate else 0, float(alpha), int(force_generic), _p(ws), SPLITK_WS_BYTES)
    else:
        call("mla_gemm_bf16", _p(a), _p(b), _p(out), _p(residual), _p(bias), M, N, K, lda, ldb, ldc, ldr or 0, a_mode, b_mode,
             out_fp32, 1 if accumulate else 0, float(alpha), int(force_generic))
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * M * N * K, (a_mode, b_mode, M, N, K)))
    return out


def gemm_kloop(mode: int = -1) -> int:
    """Selects (0 / 1) or queries (-1) the main loop of the 256x256 GEMM's k-contiguous instantiations: 1 = hand-scheduled assembly
    (default), 0 = compiler-scheduled. Same bits either way; for A/B measurements and tests."""
    return int(lib().mla_gemm_kloop(int(mode)))


def gemm_sq_slots(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor) -> int:
    """Number of sum-of-squares partials gemm_sq() writes for these operands, or -1 when the shape is outside its contract."""
    M, K = a.shape
    N = b.shape[0]
    ok = (a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16 and out.dtype == torch.float32 and b.shape[1] == K and
          a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1 and M >= 256 and N >= 256 and K % 64 == 0 and N % 8 == 0 and
          a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and out.stride(0) % 8 == 0 and
          all(t.data_ptr() % 16 == 0 for t in (a, b, out)) and tuple(out.shape) == (M, N))
    if not ok:
        return -1
    return int(lib().mla_gemm_sq_slots(M, N, K, SPLITK_WS_BYTES if SPLITK else 0))


def gemm_sq(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, accumulate: bool, part: Optional[torch.Tensor] = None):
    """out[M, N] (fp32) (+)= a[M, K] b[N, K]^T like gemm(), and sum(out^2) of the FINAL values as partial sums written to `part`
    (fp32, >= gemm_sq_slots() elements; allocated here when None): returns (part, number of valid partials) -- or None when the shape
    is outside the 256x256 kernel (the caller then runs gemm() and the gradient norm reads the buffer as before).
    mla_sum_partials adds the partials up in a fixed order."""
    need = gemm_sq_slots(a, b, out)
    if need < 0:
        return None
    M, K = a.shape
    N = b.shape[0]
    if part is None:
        part = torch.empty(need, dtype=torch.float32, device=out.device)
    assert part.dtype == torch.float32 and part.is_contiguous() and part.numel() >= need
    slots = ctypes.c_int(0)
    ws = workspace(SPLITK_WS_BYTES, a.device) if SPLITK else None
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    call("mla_gemm_bf16_ws_sq", _p(a), _p(b), _p(out), M, N, K, a.stride(0), b.stride(0), out.stride(0), 1 if accumulate else 0, 1.0,
         _p(ws), SPLITK_WS_BYTES if ws is not None else 0, _p(part), part.numel(), ctypes.byref(slots))
    assert int(slots.value) == need
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * M * N * K, (0, 0, M, N, K)))
    return part, need


def sum_partials(partials: torch.Tensor, n: int, out1: torch.Tensor, accumulate: bool):
    call("mla_sum_partials", _p(partials), int(n), _p(out1), 1 if accumulate else 0)


def _rs_args(norm, rows, device):
    """(ss, parts, eps, rstd) of a folded RMSNorm for the _rs entry points. norm = (ss [rows, parts] fp32 or None, rstd [rows] fp32 or
    None, eps): partials given -> rstd is computed by the launch and returned; else the given rstd is read."""
    ss, rstd, eps = norm
    if ss is not None:
        assert ss.dtype == torch.float32 and ss.is_contiguous() and ss.shape[0] == rows
        rstd = torch.empty(rows, dtype=torch.float32, device=device)
        return ss, int(ss.shape[1]), float(eps), rstd
    assert rstd is not None and rstd.dtype == torch.float32 and rstd.is_contiguous() and rstd.numel() == rows
    return None, 0, float(eps), rstd


def qkv_rope_ok(x2d, wqkv, out, cos, sin, S, rope_cols) -> bool:
    T, K = x2d.shape
    N = wqkv.shape[0]
    return (T >= 256 and N >= 256 and K % 64 == 0 and N % 8 == 0 and rope_cols % 256 == 0 and x2d.stride(0) % 8 == 0 and
            wqkv.stride(0) % 8 == 0 and out.stride(0) % 8 == 0 and cos.dtype == torch.float32 and cos.is_contiguous() and
            sin.is_contiguous() and cos.shape == (S, 64) and all(t.data_ptr() % 16 == 0 for t in (x2d, wqkv, out, cos, sin)))


def gemm_qkv_rope(x2d, wqkv, out, cos, sin, S, rope_cols, norm=None):
    """out[T, N] = x2d @ wqkv^T with RoPE applied to columns [0, rope_cols) in the GEMM epilogue (bit-identical to gemm + rope_inplace).
    Returns False when the shape is outside the fused kernel's contract (the caller then runs the two separate launches).
    norm = (ss, rstd, eps): x2d is x * g of a folded RMSNorm (mla_gemm_qkv_rope_rs: fused RMSNorm + QKV + RoPE); returns rstd [T]."""
    T, K = x2d.shape
    N = wqkv.shape[0]
    ok = qkv_rope_ok(x2d, wqkv, out, cos, sin, S, rope_cols)
    if not ok:
        return False
    if norm is not None:
        _req(x2d, torch.bfloat16, "gemm_qkv_rope x")
        _req(wqkv, torch.bfloat16, "gemm_qkv_rope w")
        ss, parts, eps, rstd = _rs_args(norm, T, x2d.device)
        prof = GEMM_PROFILE
        if prof is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        call("mla_gemm_qkv_rope_rs", _p(x2d), _p(wqkv), _p(out), T, N, K, x2d.stride(0), wqkv.stride(0), out.stride(0), _p(cos), _p(sin),
             int(S), int(rope_cols), _p(ss), parts, eps, _p(rstd))
        if prof is not None:
            ev1.record()
            prof.append((ev0, ev1, 2.0 * T * N * K, (0, 0, T, N, K, "rope_epilogue")))
        return rstd
    _req(x2d, torch.bfloat16, "gemm_qkv_rope x")
    _req(wqkv, torch.bfloat16, "gemm_qkv_rope w")
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    call("mla_gemm_qkv_rope", _p(x2d), _p(wqkv), _p(out), T, N, K, x2d.stride(0), wqkv.stride(0), out.stride(0), _p(cos), _p(sin),
         int(S), int(rope_cols))
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * T * N * K, (0, 0, T, N, K, "rope_epilogue")))
    return True


def gateup_swiglu_ok(x2d, wgu, want_t) -> bool:
    T, K = x2d.shape
    I = wgu.shape[0] // 2
    return (T >= 256 and I % 128 == 0 and K % 64 == 0 and wgu.shape[0] == 2 * I and x2d.stride(0) % 8 == 0 and wgu.stride(0) % 8 == 0 and
            (not want_t or T % 8 == 0) and all(t.data_ptr() % 16 == 0 for t in (x2d, wgu)))


def gemm_gateup_swiglu(x2d, wgu, want_t, norm=None, want_act=True, want_gu=True):
    """(gu [T, 2I], act [T, I], actT [I, T] or None) from ONE launch: the SwiGLU product is formed in the gate|up GEMM's epilogue.
    None when the shape is outside the fused kernel's contract (the caller then runs gemm + swiglu_fwd[_dual]).
    norm = (ss, rstd, eps): x2d is x * g of a folded RMSNorm (mla_gemm_gateup_swiglu_rs); rstd [T] is appended to the result."""
    T, K = x2d.shape
    I = wgu.shape[0] // 2
    ok = gateup_swiglu_ok(x2d, wgu, want_t)
    if not ok:
        return None
    _req(x2d, torch.bfloat16, "gemm_gateup_swiglu x")
    _req(wgu, torch.bfloat16, "gemm_gateup_swiglu w")
    act = torch.empty((T, I), dtype=torch.bfloat16, device=x2d.device) if (want_act or not want_t) else None   # want_act=False: act^T only
    gu = torch.empty((T, 2 * I), dtype=torch.bfloat16, device=x2d.device) if (want_gu or act is None) else None  # want_gu=False: the product only
    actT = torch.empty((I, T), dtype=torch.bfloat16, device=x2d.device) if want_t else None
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rstd = None
    if norm is not None:
        ss, parts, eps, rstd = _rs_args(norm, T, x2d.device)
        call("mla_gemm_gateup_swiglu_rs", _p(x2d), _p(wgu), _p(gu), _p(act), _p(actT), T, I, K, x2d.stride(0), wgu.stride(0), T,
             _p(ss), parts, eps, _p(rstd))
    else:
        call("mla_gemm_gateup_swiglu", _p(x2d), _p(wgu), _p(gu), _p(act), _p(actT), T, I, K, x2d.stride(0), wgu.stride(0), T)
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * T * 2 * I * K, (0, 0, T, 2 * I, K, "swiglu_fwd_epilogue")))
    return (gu, act, actT) if norm is None else (gu, act, actT, rstd)


def gemm_dact_swiglu_bwd(dy2d, wT, gu2d):
    """(dgu [T, 2I], dguT [2I, T]) = SwiGLU backward of d(act) = dy2d @ wT^T without ever writing d(act); None when the shape is outside
    the fused kernel's contract (the caller then runs gemm + swiglu_bwd_t)."""
    T, K = dy2d.shape
    I = wT.shape[0]
    ok = (T >= 256 and I >= 256 and K % 64 == 0 and I % 8 == 0 and T % 8 == 0 and gu2d.shape == (T, 2 * I) and gu2d.is_contiguous() and
          dy2d.stride(0) % 8 == 0 and wT.stride(0) % 8 == 0 and all(t.data_ptr() % 16 == 0 for t in (dy2d, wT, gu2d)))
    if not ok:
        return None
    _req(dy2d, torch.bfloat16, "gemm_dact_swiglu_bwd dy")
    _req(wT, torch.bfloat16, "gemm_dact_swiglu_bwd wT")
    dgu = torch.empty_like(gu2d)
    dguT = torch.empty((2 * I, T), dtype=torch.bfloat16, device=gu2d.device)
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    call("mla_gemm_dact_swiglu_bwd", _p(dy2d), _p(wT), _p(gu2d), _p(dgu), _p(dguT), T, I, K, dy2d.stride(0), wT.stride(0), T)
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * T * I * K, (0, 0, T, I, K, "swiglu_bwd_epilogue")))
    return dgu, dguT


# --------------------------------------------------------------------------------------------- norms
def rmsnorm_fwd(x2d, w, eps):
    _req(x2d, torch.bfloat16, "rmsnorm x")
    _req(w, torch.bfloat16, "rmsnorm w")
    rows, H = x2d.shape
    y = torch.empty_like(x2d)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    call("mla_rmsnorm_fwd", _p(x2d), _p(w), _p(y), _p(rstd), rows, H, float(eps))
    return y, rstd


def rmsnorm_prep(x2d, w, eps, want_rstd=True):
    """(xg = bf16(x * w), rstd [rows] or None): the two halves of a folded RMSNorm for rows that do not come out of a GEMM epilogue."""
    _req(x2d, torch.bfloat16, "rmsnorm_prep x")
    _req(w, torch.bfloat16, "rmsnorm_prep w")
    rows, H = x2d.shape
    xg = torch.empty_like(x2d)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device) if want_rstd else None
    call("mla_rmsnorm_prep", _p(x2d), _p(w), _p(xg), _p(rstd), rows, H, float(eps))
    return xg, rstd


def res_norm_ok(a, b, residual) -> bool:
    M, K = a.shape
    N = b.shape[0]
    return (M >= 256 and N >= 256 and N % 256 == 0 and K % 64 == 0 and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and
            residual.stride(0) % 8 == 0 and a.stride(1) == 1 and b.stride(1) == 1 and residual.stride(1) == 1 and
            all(t.data_ptr() % 16 == 0 for t in (a, b, residual)))


def gemm_res_norm(a, b, residual, g):
    """(h, xg, ss): h = a @ b^T + residual (the new residual-stream rows), xg = bf16(h * g) and ss [M, N / 256] = per-tile partials of
    sum(h^2) per row -- the producer half of a folded RMSNorm (mla_gemm_res_norm), one GEMM launch."""
    _req(a, torch.bfloat16, "gemm_res_norm a")
    _req(b, torch.bfloat16, "gemm_res_norm b")
    _req(residual, torch.bfloat16, "gemm_res_norm residual")
    _req(g, torch.bfloat16, "gemm_res_norm g")
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and residual.shape == (M, N) and g.numel() == N and g.data_ptr() % 16 == 0
    h = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    xg = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    ss = torch.empty((M, N // 256), dtype=torch.float32, device=a.device)
    ws = workspace(SPLITK_WS_BYTES, a.device) if SPLITK else None
    prof = GEMM_PROFILE
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    call("mla_gemm_res_norm", _p(a), _p(b), _p(h), _p(residual), _p(g), _p(xg), _p(ss), M, N, K, a.stride(0), b.stride(0), N,
         residual.stride(0), _p(ws), SPLITK_WS_BYTES if ws is not None else 0)
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1, 2.0 * M * N * K, (0, 0, M, N, K)))
    return h, xg, ss


def rmsnorm_bwd(dy, x2d, w, rstd, dres=None, dw_out=None, dw_accumulate=False):
    rows, H = x2d.shape
    dx = torch.empty_like(x2d)
    nb = lib().mla_rmsnorm_bwd_blocks(rows)
    ws = workspace(nb * H * 4, x2d.device) if dw_out is not None else None
    call("mla_rmsnorm_bwd", _p(dy), _p(x2d), _p(w), _p(rstd), _p(dres), _p(dx), _p(dw_out), 1 if dw_accumulate else 0, rows, H,
         _p(ws), ws.numel() if ws is not None else 0)
    return dx


def timm_rmsnorm_fwd(x2d, w, eps):
    """timm==0.9.10 RmsNorm (torch.var based, see include/mla_hip.h). Returns (y, mean, rstd)."""
    _req(x2d, torch.bfloat16, "timm_rmsnorm x")
    _req(w, torch.bfloat16, "timm_rmsnorm w")
    rows, H = x2d.shape
    y = torch.empty_like(x2d)
    stats = torch.empty(2, rows, dtype=torch.float32, device=x2d.device)
    call("mla_timm_rmsnorm_fwd", _p(x2d), _p(w), _p(y), _p(stats[0]), _p(stats[1]), rows, H, float(eps))
    return y, stats[0], stats[1]


def timm_rmsnorm_bwd(dy, x2d, w, mean, rstd, dw_out=None, dw_accumulate=False):
    rows, H = x2d.shape
    dx = torch.empty_like(x2d)
    nb = lib().mla_rmsnorm_bwd_blocks(rows)
    ws = workspace(nb * H * 4, x2d.device) if dw_out is not None else None
    call("mla_timm_rmsnorm_bwd", _p(dy), _p(x2d), _p(w), _p(mean), _p(rstd), _p(dx), _p(dw_out), 1 if dw_accumulate else 0, rows, H,
         _p(ws), ws.numel() if ws is not None else 0)
    return dx


def colsum(dy2d, out_f32, accumulate):
    rows, N = dy2d.shape
    rs = lib().mla_colsum_blocks(rows)
    ws = workspace(rs * N * 4, dy2d.device)
    call("mla_colsum_bf16", _p(dy2d), _p(out_f32), 1 if accumulate else 0, rows, N, dy2d.stride(0), _p(ws), ws.numel())


def layernorm_fwd(x2d, w, b, eps):
    rows, H = x2d.shape
    y = torch.empty_like(x2d)
    call("mla_layernorm_fwd", _p(x2d), _p(w), _p(b), _p(y), rows, H, float(eps))
    return y


def rope_inplace(buf2d, cos, sin, S, nheads, D, q_off, k_off, backward=False):
    tokens = buf2d.shape[0]
    call("mla_rope_inplace", _p(buf2d), _p(cos), _p(sin), tokens, S, nheads, D, buf2d.stride(0), q_off, k_off,
         1 if backward else 0)


def swiglu_fwd(gu2d):
    rows, two_i = gu2d.shape
    act = torch.empty((rows, two_i // 2), dtype=torch.bfloat16, device=gu2d.device)
    call("mla_swiglu_fwd", _p(gu2d), _p(act), rows, two_i // 2)
    return act


def swiglu_bwd(dact, gu2d, want_act=False):
    rows, two_i = gu2d.shape
    dgu = torch.empty_like(gu2d)
    act = torch.empty_like(dact) if want_act else None
    call("mla_swiglu_bwd", _p(dact), _p(gu2d), _p(dgu), _p(act), rows, two_i // 2)
    return dgu, act


def swiglu_bwd_t(dact, gu2d):
    """dgu [rows, 2I] and its transpose [2I, rows] from one pass over dact and gu."""
    rows, two_i = gu2d.shape
    dgu = torch.empty_like(gu2d)
    dguT = torch.empty((two_i, rows), dtype=gu2d.dtype, device=gu2d.device)
    call("mla_swiglu_bwd_t", _p(dact), _p(gu2d), _p(dgu), _p(dguT), rows, two_i // 2, rows)
    return dgu, dguT


def act_fwd(x, kind):
    y = torch.empty_like(x)
    call("mla_act_fwd", _p(x), _p(y), x.numel(), kind)
    return y


def act_bwd(dy, x, kind):
    dx = torch.empty_like(x)
    call("mla_act_bwd", _p(dy), _p(x), _p(dx), x.numel(), kind)
    return dx


def cast_f32_to_bf16(x, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    call("mla_cast_f32_to_bf16", _p(x), _p(out), x.numel())
    return out


def cast_bf16_to_f32(x, out=None):
    if out is None:
        out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    call("mla_cast_bf16_to_f32", _p(x), _p(out), x.numel())
    return out


def add_bf16(a, b):
    y = torch.empty_like(a)
    call("mla_add_bf16", _p(a), _p(b), _p(y), a.numel())
    return y


def embedding_fwd(ids, table):
    tokens = ids.numel()
    vocab, H = table.shape
    out = torch.empty((tokens, H), dtype=torch.bfloat16, device=table.device)
    call("mla_embedding_fwd", _p(ids), _p(table), _p(out), tokens, H, vocab)
    return out


def embedding_bwd(ids, dy2d, grad_f32):
    """grad_f32[ids[t]] += dy2d[t], deterministic (ascending token order; an aligned batch of 64 equal ids -- padding -- is summed
    first into the workspace row the library asks for)."""
    vocab, H = grad_f32.shape
    tokens = ids.numel()
    ws = torch.empty((tokens // 64, H), dtype=torch.float32, device=grad_f32.device) if tokens >= 64 else None
    call("mla_embedding_bwd", _p(ids), _p(dy2d), _p(grad_f32), _p(ws), tokens, H, vocab)


def adamw_step(p32, g32, m, v, p16, lr, beta1, beta2, eps, wd, step, grad_scale=None):
    call("mla_adamw_step", _p(p32), _p(g32), _p(m), _p(v), _p(p16), p32.numel(), float(lr), float(beta1), float(beta2),
         float(eps), float(wd), int(step), _p(grad_scale))


def adamw_step_groups(p32, g32, m, v, p16, n_decay, lr, beta1, beta2, eps, wd, step, grad_scale=None):
    """adamw_step over a flat range laid out [decayed | not decayed]: weight decay on the first n_decay elements only."""
    call("mla_adamw_step_groups", _p(p32), _p(g32), _p(m), _p(v), _p(p16), p32.numel(), int(n_decay), float(lr), float(beta1), float(beta2),
         float(eps), float(wd), int(step), _p(grad_scale))


def sumsq(x32, out1, accumulate):
    ws = workspace(8192, x32.device)
    call("mla_sumsq_f32", _p(x32), x32.numel(), _p(out1), 1 if accumulate else 0, _p(ws), ws.numel())


def clip_coef(sumsq1, max_norm, coef1, norm1=None):
    call("mla_clip_coef", _p(sumsq1), float(max_norm), _p(coef1), _p(norm1))


def q_sample(x0, noise, t, sqrt_ac, sqrt_1mac):
    out = torch.empty_like(x0)
    batch = x0.shape[0]
    call("mla_q_sample", _p(x0), _p(noise), _p(t), _p(sqrt_ac), _p(sqrt_1mac), _p(out), batch, x0.numel() // batch,
         sqrt_ac.numel())
    return out


# --------------------------------------------------------------------------------------------- attention
def _group_args(groups, B):
    """groups = (start, len): start an int (every sample) or an int32 device tensor [B] (per sample) -> (scalar start, tensor or None)"""
    start = groups[0]
    if torch.is_tensor(start):
        _req(start, torch.int32, "attention group starts")
        assert start.numel() == B and start.is_contiguous()
        return 0, start
    return int(start), None


def attn_fwd(q, k, v, B, S, H, D, ld_qkv, seqlens, scale, rows=None, groups=None):
    """q/k/v: views into the packed [B*S, 3*H*D] buffer (first element of each slice). rows > B*S: the output gets that many rows,
    the extra ones zero (row padding of the caller, ops.DecoderLayerFn). groups = (start, len): shared-prefix sequences -- rows >= start
    are suffix groups of `len` rows, a query sees the prefix and (causally) its own group only (mla_attn_fwd_g)."""
    rows = B * S if rows is None else rows
    o = torch.empty((rows, H * D), dtype=torch.bfloat16, device=q.device)
    if rows > B * S:
        o[B * S:].zero_()
    lse = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
    if groups is not None:
        gs, gst = _group_args(groups, B)
        call("mla_attn_fwd_g", _p(q), _p(k), _p(v), _p(o), _p(lse), _p(seqlens), B, S, H, D, ld_qkv, H * D, float(scale), gs, int(groups[1]), _p(gst))
    else:
        call("mla_attn_fwd", _p(q), _p(k), _p(v), _p(o), _p(lse), _p(seqlens), B, S, H, D, ld_qkv, H * D, float(scale))
    return o, lse


# The five-product backward is OPT-IN (MLA_ATTN_BWD5=1): measured on MI355X it is not faster than the two-kernel, seven-product form
# (S = 548: delta 81 + dK/dV 438 + dQ 262 us vs 419 + 352; S = 2048: 189 + 3420 + 1681 vs 3039 + 2218) -- MFMA work is cheap here
# and the one-product dQ kernel is bound by streaming dS^T and re-streaming K through LDS.
ATTN_BWD5 = os.environ.get("MLA_ATTN_BWD5", "0") == "1"


# ---- one-launch backward: the head counters are OURS (the library keeps nothing), one zero-initialised int32 buffer per (device, stream),
# handed over only on a device where the dispatch probe holds (mla_hip.h: mla_attn_bwd, mla_dispatch_probe).
_HEAD_SYNC = {}          # (device index, stream handle) -> int32 tensor; the kernel leaves it zero after every launch
_DISPATCH_OK = {}        # device index -> (bool, dict): result of dispatch_probe()
ATTN_BWD_MERGED = os.environ.get("MLA_ATTN_BWD_MERGED", "105") != "0"


def dispatch_probe(device="cuda", blocks=4096, hold_us=20):
    """Measures what attn_bwd_merged_kernel assumes (include/mla_hip.h: mla_dispatch_probe). Returns (ok, info): ok iff there are 8
    XCDs, workgroup L ran on XCD L & 7 for every L, and per XCD no workgroup took its start ticket more than two residency rounds (2 x 64
    slots: 32 CUs x 2 workgroups) away from its place in id order -- the workgroups of one round start together and take their tickets
    in any order (measured on MI355X: worst displacement 52-56), an out-of-order dispatcher would show displacements of the grid's
    size. Synchronises -- called once per device, outside any capture."""
    dev = torch.device(device)
    out = torch.zeros(1 + 2 * blocks, dtype=torch.int32, device=dev)
    call("mla_dispatch_probe", _p(out), blocks, hold_us)
    rec = out.cpu()[1:].view(blocks, 2)
    ticket, xcc = rec[:, 0].long(), rec[:, 1].long()
    ids = torch.arange(blocks)
    n_xcd = int(xcc.unique().numel())
    xcd_ok = bool((xcc == (ids & 7)).all())
    worst = 0
    for x in range(8):
        t = ticket[ids % 8 == x]                       # start tickets of the workgroups with id = x mod 8, in id order
        rank = torch.argsort(torch.argsort(t))          # place of each in start order
        worst = max(worst, int((rank - torch.arange(t.numel())).abs().max()))
    info = {"xcds": n_xcd, "xcc_is_id_mod_8": xcd_ok, "worst_start_displacement": worst, "blocks": blocks,
            "tickets_complete": bool(torch.equal(torch.sort(ticket).values, ids))}
    return (n_xcd == 8 and xcd_ok and worst <= 128 and info["tickets_complete"]), info


def _head_sync(device, B, H):
    """The caller-owned counter pair per head for the one-launch backward, or None (= two launches)."""
    if not ATTN_BWD_MERGED:
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _DISPATCH_OK:
        if torch.cuda.is_current_stream_capturing():
            return None                                  # first use inside a capture: the probe synchronises -- stay on two launches
        with torch.cuda.device(idx):
            _DISPATCH_OK[idx] = dispatch_probe(torch.device("cuda", idx))
    if not _DISPATCH_OK[idx][0]:
        return None
    need = int(lib().mla_attn_bwd_sync_ints(B, H))
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    buf = _HEAD_SYNC.get(key)
    if buf is None or buf.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            return None
        # (an outgrown buffer may still be read by queued launches: torch's stream-ordered allocator does not hand its memory to
        # anything that could run before them)
        buf = torch.zeros(max(need, 4096), dtype=torch.int32, device=torch.device("cuda", idx))
        _HEAD_SYNC[key] = buf
    return buf


def attn_bwd(q, k, v, o, dout, lse, seqlens, dq, dk, dv, B, S, H, D, ld_qkv, scale, rope_cos=None, rope_sin=None, transposed=None,
             five: Optional[bool] = None, merged: Optional[bool] = None, groups=None):
    """five (experiment build only, see ATTN_BWD5): the five-product form -- the dK / dV kernel hands dS^T (bf16 tiles in a torch-owned
    scratch buffer) to a one-product dQ kernel instead of both kernels recomputing Q K^T and dO V^T (mla_attn_bwd_ws).
    merged (default: on where the dispatch probe holds; MLA_ATTN_BWD_MERGED=0 turns it off): one launch for the dQ and dK / dV blocks
    with this module's per-stream head counters; False = two launches. Bit-identical either way.
    rope_cos / rope_sin ([S, D/2] fp32): dq / dk come out with the RoPE backward already applied (no separate pass).
    transposed = (dqkvT [3*H*D, ldt], oT [H*D, ldt]) bf16: also filled with the token-contiguous copies of dq|dk|dv and o (columns
    b*S + s; columns >= B*S are left alone) -- the wgrad operands, without transpose passes. Needs S % 4 == 0, ldt % 4 == 0."""
    delta = torch.empty((B, H, S), dtype=torch.float32, device=q.device)
    five = ATTN_BWD5 if five is None else five
    if rope_cos is not None:
        _req(rope_cos, torch.float32, "rope cos")
        _req(rope_sin, torch.float32, "rope sin")
        assert rope_cos.shape in ((S, D // 2), (B * S, D // 2)) and rope_sin.shape == rope_cos.shape and rope_cos.is_contiguous() and rope_sin.is_contiguous()
        assert rope_cos.shape[0] == S or groups is not None, "per-sample RoPE tables [B * S, 64] go with mla_attn_bwd_g (groups=...)"
    if transposed is not None:
        dqkvT, oT = transposed
        _req(dqkvT, torch.bfloat16, "dqkvT")
        _req(oT, torch.bfloat16, "oT")
        ldt = dqkvT.stride(0)
        assert dqkvT.shape[0] == 3 * H * D and oT.shape[0] == H * D and oT.stride(0) == ldt and dqkvT.stride(1) == 1 and oT.stride(1) == 1
    if five:
        nbytes = int(lib().mla_attn_bwd_ws_bytes(B, S, H))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)           # transient: freed (to torch's pool) after the launches are queued
        if transposed is not None:
            tq, tk, tv, to_ = dqkvT[:H * D], dqkvT[H * D:2 * H * D], dqkvT[2 * H * D:], oT
        else:
            tq = tk = tv = to_ = None
            ldt = 0
        call("mla_attn_bwd_ws", _p(q), _p(k), _p(v), _p(o), _p(dout), _p(lse), _p(seqlens), _p(dq), _p(dk), _p(dv), _p(delta), B, S,
             H, D, ld_qkv, H * D, float(scale), _p(rope_cos), _p(rope_sin), _p(tq), _p(tk), _p(tv), _p(to_), ldt, _p(ws), nbytes)
        return
    sync = _head_sync(q.device, B, H) if merged is None or merged else None
    if merged and sync is None:
        raise RuntimeError("attn_bwd(merged=True): the one-launch backward is not available here "
                           f"(MLA_ATTN_BWD_MERGED=0, a stream capture before the first use, or the dispatch probe failed: {_DISPATCH_OK})")
    n_sync = sync.numel() if sync is not None else 0
    if groups is not None:                               # shared-prefix sequences (mla_attn_bwd_g; see attn_fwd)
        tq = tk = tv = to_ = None
        if transposed is not None:
            tq, tk, tv, to_ = dqkvT[:H * D], dqkvT[H * D:2 * H * D], dqkvT[2 * H * D:], oT
        gs, gst = _group_args(groups, B)
        call("mla_attn_bwd_g", _p(q), _p(k), _p(v), _p(o), _p(dout), _p(lse), _p(seqlens), _p(dq), _p(dk), _p(dv), _p(delta), B, S,
             H, D, ld_qkv, H * D, float(scale), _p(rope_cos), _p(rope_sin), _p(tq), _p(tk), _p(tv), _p(to_), ldt if transposed is not None else 0,
             _p(sync), n_sync, gs, int(groups[1]), _p(gst), 1 if (rope_cos is not None and rope_cos.shape[0] == B * S and B > 1) else 0)
        return
    if transposed is not None:
        call("mla_attn_bwd_t", _p(q), _p(k), _p(v), _p(o), _p(dout), _p(lse), _p(seqlens), _p(dq), _p(dk), _p(dv), _p(delta), B, S,
             H, D, ld_qkv, H * D, float(scale), _p(rope_cos), _p(rope_sin), _p(dqkvT[:H * D]), _p(dqkvT[H * D:2 * H * D]),
             _p(dqkvT[2 * H * D:]), _p(oT), ldt, _p(sync), n_sync)
        return
    call("mla_attn_bwd", _p(q), _p(k), _p(v), _p(o), _p(dout), _p(lse), _p(seqlens), _p(dq), _p(dk), _p(dv), _p(delta), B, S,
         H, D, ld_qkv, H * D, float(scale), _p(rope_cos), _p(rope_sin), _p(sync), n_sync)


# --------------------------------------------------------------------------------------------- inference (mla_amd/infer.py)
def gemv(x, W, out, ldo, out_batch_stride, rows_per_batch, residual=None, out_col=0, norm_weight=None, eps=0.0, swiglu=False, rope=None):
    """out row m (at out + (m // rows_per_batch) * out_batch_stride + (m % rows_per_batch) * ldo + out_col) = f(x[m]) @ W^T (+ residual[m]);
    M <= 8 rows, every weight row read once (mla_gemv_bf16). `out` is a base tensor: only its data pointer is used.
    f = identity; or LlamaRMSNorm(x; norm_weight, eps); or (swiglu=True, x = packed gate|up rows [M, 2 K]) silu(gate) * up.
    rope = (cos [rows_per_batch, 64], sin, rope_cols): output columns [0, rope_cols) are rotated per head of 128 in the epilogue."""
    _req(x, torch.bfloat16, "gemv x")
    _req(W, torch.bfloat16, "gemv W")
    _req(out, torch.bfloat16, "gemv out")
    M = x.shape[0]
    K = x.shape[1] // 2 if swiglu else x.shape[1]
    N = W.shape[0]
    assert W.shape[1] == K and x.stride(1) == 1 and W.stride(1) == 1 and not (swiglu and norm_weight is not None)
    pre = 2 if swiglu else (1 if norm_weight is not None else 0)
    if norm_weight is not None:
        _req(norm_weight, torch.bfloat16, "gemv norm weight")
        assert norm_weight.numel() == K and norm_weight.is_contiguous()
    if rope is not None:
        _req(rope[0], torch.float32, "gemv rope cos")
        _req(rope[1], torch.float32, "gemv rope sin")
        assert rope[0].shape == (rows_per_batch, 64) and rope[1].shape == rope[0].shape and rope[0].is_contiguous() and rope[1].is_contiguous()
        assert out_col == 0 and residual is None
    if residual is not None:
        _req(residual, torch.bfloat16, "gemv residual")
        assert residual.shape[0] == M and residual.stride(1) == 1
    call("mla_gemv_bf16", _p(x), x.stride(0), _p(W), W.stride(0), c_void_p(out.data_ptr() + 2 * out_col), ldo, out_batch_stride, rows_per_batch,
         _p(residual), residual.stride(0) if residual is not None else 0, M, N, K, pre, _p(norm_weight), float(eps),
         _p(rope[0]) if rope is not None else None, _p(rope[1]) if rope is not None else None, int(rope[2]) if rope is not None else 0)


def attn_decode(cache, B, nheads, D, S_kv, R, scale):
    """cache [B, S_cap >= S_kv, 3 * nheads * D] packed post-RoPE q|k|v rows; the R query rows are rows [S_kv - R, S_kv) of every sample,
    query r attends to keys [0, S_kv - R + r] (mla_attn_decode). Returns o [B * R, nheads * D] bf16."""
    _req(cache, torch.bfloat16, "attn_decode cache")
    H = nheads * D
    assert cache.shape[0] == B and cache.shape[2] == 3 * H and cache.stride(2) == 1 and cache.shape[1] >= S_kv
    o = torch.empty((B * R, H), dtype=torch.bfloat16, device=cache.device)
    base = cache.data_ptr()
    call("mla_attn_decode", c_void_p(base), c_void_p(base + 2 * H), c_void_p(base + 4 * H), _p(o), B, nheads, D, S_kv, R, cache.stride(1),
         cache.stride(0), H, float(scale))
    return o


# --------------------------------------------------------------------------------------------- losses
def ce_fwd(logits2d, labels, ncols=None, ignore_index=-100, want_loss=True):
    rows = logits2d.shape[0]
    ncols = logits2d.shape[1] if ncols is None else ncols
    loss = torch.empty(rows, dtype=torch.float32, device=logits2d.device) if want_loss else None
    lse = torch.empty(rows, dtype=torch.float32, device=logits2d.device)
    call("mla_ce_fwd", _p(logits2d), 1 if logits2d.dtype == torch.float32 else 0, logits2d.stride(0), _p(labels), _p(loss),
         _p(lse), rows, ncols, ignore_index)
    return loss, lse


def ce_bwd(logits2d, labels, lse, gscale1, inv_count, ncols=None, ignore_index=-100):
    rows = logits2d.shape[0]
    ncols = logits2d.shape[1] if ncols is None else ncols
    d = torch.zeros(logits2d.shape, dtype=torch.bfloat16, device=logits2d.device)
    call("mla_ce_bwd", _p(logits2d), 1 if logits2d.dtype == torch.float32 else 0, logits2d.stride(0), _p(labels), _p(lse),
         _p(gscale1), float(inv_count), _p(d), d.stride(0), rows, ncols, ignore_index)
    return d


def infonce_bwd(L, rlse, clse, gscale1, M):
    Mp = L.shape[0]
    dL = torch.empty((Mp, Mp), dtype=torch.bfloat16, device=L.device)
    call("mla_infonce_bwd", _p(L), _p(rlse), _p(clse), _p(gscale1), _p(dL), M, Mp)
    return dL


def l2norm_fwd(x2d, eps=1e-12):
    rows, n = x2d.shape
    y = torch.empty_like(x2d)
    norms = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    call("mla_l2norm_fwd", _p(x2d), _p(y), _p(norms), rows, n, float(eps))
    return y, norms


def l2norm_bwd(dy, y, norms):
    rows, n = y.shape
    dx = torch.empty_like(y)
    call("mla_l2norm_bwd", _p(dy), _p(y), _p(norms), _p(dx), rows, n)
    return dx


def calib_mfma(device="cuda", total_ms=300.0, measure_ms=200.0, blocks=None):
    """The random-operand MFMA stream of this box (include/mla_hip.h: mla_calib_mfma): launches of ~20 ms back to back for ~total_ms,
    the rate over the last ~measure_ms (HIP events on the launch stream) in PFLOP/s -- the first launches ramp the power controller.
    Synchronises. Returns dict(pflops, launches, ms_per_launch)."""
    dev = torch.device(device)
    if blocks is None:
        blocks = torch.cuda.get_device_properties(dev).multi_processor_count
    ops_ = torch.randn(512 * 24 * 8, device=dev).to(torch.bfloat16)
    out = torch.empty(blocks * 512, dtype=torch.float32, device=dev)
    fl = ctypes.c_double(0.0)
    iters = 16384                                          # ~19 ms per launch at 1.9 PFLOP/s on 256 CUs
    call("mla_calib_mfma", _p(ops_), _p(out), blocks, 256, ctypes.byref(fl))          # code-object load + clocks up
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call("mla_calib_mfma", _p(ops_), _p(out), blocks, iters, ctypes.byref(fl))
    e1.record()
    torch.cuda.synchronize(dev)
    one = max(e0.elapsed_time(e1), 1e-3)
    n_total = max(2, int(round(total_ms / one)))
    n_meas = max(1, min(n_total - 1, int(round(measure_ms / one))))
    evs = []
    for i in range(n_total):
        if i == n_total - n_meas:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            evs.append(ev)
        call("mla_calib_mfma", _p(ops_), _p(out), blocks, iters, ctypes.byref(fl))
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    evs.append(ev)
    torch.cuda.synchronize(dev)
    ms = evs[0].elapsed_time(evs[1])
    return {"pflops": fl.value * n_meas / (ms * 1e-3) / 1e15, "launches": n_meas, "ms_per_launch": ms / n_meas, "blocks": blocks}


def selftest(device="cuda"):
    src = torch.arange(256, dtype=torch.int32, device=device) * 7 + 3
    out_tr = torch.zeros(256, dtype=torch.int32, device=device)
    out_g = torch.zeros(256, dtype=torch.int32, device=device)
    call("mla_selftest", _p(src), _p(out_tr), _p(out_g))
    return src, out_tr, out_g


# --------------------------------------------------------------------------------------------- point cloud / vision
register_signatures({
    "mla_project_points": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_int, c_int, c_void_p],
    "mla_fps": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mla_knn": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "mla_lga_prep": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                     c_float, c_void_p],
    "mla_colstats_bf16": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p, c_size_t, c_void_p],
    "mla_bn_apply": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_float, c_int,
                     c_void_p],
    "mla_maxpool_k": [c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p],
    "mla_gather_rows_f32": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    "mla_im2col_patch": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "mla_avgpool_tokens": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "mla_local_attn": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "mla_local_attn_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p],
    "mla_swiglu_bwd_t": [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_longlong, c_void_p],
    "mla_lga_prep_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "mla_maxpool_k_bwd": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p],
    "mla_clip_preprocess": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                            c_int, c_void_p, c_void_p, c_int, c_void_p],
    "mla_avgpool_tokens_bwd": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "mla_colstats_blocks": [c_longlong],
    "mla_gather_rows_bf16": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p],
    "mla_transpose_bf16": [c_void_p, c_void_p, c_longlong, c_int, c_longlong, c_longlong, c_void_p],
    "mla_rmsnorm_apply_t": [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_longlong, c_void_p],
    "mla_swiglu_fwd_t": [c_void_p, c_void_p, c_longlong, c_int, c_longlong, c_void_p],
    "mla_swiglu_fwd_dual": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_longlong, c_void_p],
})


def transpose(x2d, out=None):
    """out[c][r] = x2d[r][c] (bf16, unit inner stride, any row stride)."""
    R, C = x2d.shape
    if out is None:
        out = torch.empty((C, R), dtype=torch.bfloat16, device=x2d.device)
    call("mla_transpose_bf16", _p(x2d), _p(out), R, C, x2d.stride(0), out.stride(0))
    return out


def rmsnorm_apply_t(x2d, w, rstd):
    """[H, T] = transpose(rmsnorm(x) with the saved rstd) -- recompute + transpose in one pass."""
    rows, H = x2d.shape
    out = torch.empty((H, rows), dtype=torch.bfloat16, device=x2d.device)
    call("mla_rmsnorm_apply_t", _p(x2d), _p(w), _p(rstd), _p(out), rows, H, rows)
    return out


def swiglu_fwd_dual(gu2d):
    """(act [rows, I], actT [I, rows]) in one pass over gu."""
    rows, two_i = gu2d.shape
    act = torch.empty((rows, two_i // 2), dtype=torch.bfloat16, device=gu2d.device)
    actT = torch.empty((two_i // 2, rows), dtype=torch.bfloat16, device=gu2d.device)
    call("mla_swiglu_fwd_dual", _p(gu2d), _p(act), _p(actT), rows, two_i // 2, rows)
    return act, actT


def swiglu_fwd_t(gu2d):
    rows, two_i = gu2d.shape
    out = torch.empty((two_i // 2, rows), dtype=torch.bfloat16, device=gu2d.device)
    call("mla_swiglu_fwd_t", _p(gu2d), _p(out), rows, two_i // 2, rows)
    return out


def gather_rows(src2d, idx, out_rows=None, scatter=False):
    """gather: out[r] = src[idx[r]] (out has len(idx) rows, or out_rows >= len(idx) rows with a zero tail);
    scatter: out[idx[r]] = src[r] (out has out_rows rows, zero-filled)."""
    H = src2d.shape[1]
    n = idx.numel()
    if scatter:
        out = torch.zeros((out_rows, H), dtype=torch.bfloat16, device=src2d.device)
    elif out_rows is not None and out_rows != n:
        assert out_rows >= n, f"gather_rows: out_rows={out_rows} < len(idx)={n} (the kernel writes len(idx) rows)"
        out = torch.zeros((out_rows, H), dtype=torch.bfloat16, device=src2d.device)
    else:
        out = torch.empty((n, H), dtype=torch.bfloat16, device=src2d.device)
    call("mla_gather_rows_bf16", _p(src2d), _p(idx), _p(out), n, H, 1 if scatter else 0)
    return out


def project_points(xyz_n3, consts21, idx_out, valid_out, W, Hh, stride, ph, pw):
    call("mla_project_points", _p(xyz_n3), _p(consts21), _p(idx_out), _p(valid_out), xyz_n3.shape[0], W, Hh, stride, ph, pw)


def fps(xyz, start, npoint):
    B, N, _ = xyz.shape
    out = torch.empty((B, npoint), dtype=torch.int64, device=xyz.device)
    call("mla_fps", _p(xyz), _p(start), _p(out), B, N, npoint)
    return out


def knn(xyz, centers, k):
    B, N, _ = xyz.shape
    G = centers.shape[1]
    out = torch.empty((B, G, k), dtype=torch.int32, device=xyz.device)
    call("mla_knn", _p(xyz), _p(centers), _p(out), B, N, G, k)
    return out


def gather_rows_f32(src_bnw, idx_bg):
    B, N, W = src_bnw.shape
    G = idx_bg.shape[1]
    out = torch.empty((B, G, W), dtype=torch.float32, device=src_bnw.device)
    call("mla_gather_rows_f32", _p(src_bnw), _p(idx_bg), _p(out), B, N, G, W)
    return out


def lga_prep(xyz, feats_bnc, fps_idx, knn_idx, alpha, beta):
    B, N, _ = xyz.shape
    C = feats_bnc.shape[2]
    G, K = knn_idx.shape[1], knn_idx.shape[2]
    rows = torch.empty((B * G * K, 2 * C), dtype=torch.bfloat16, device=xyz.device)
    lc = torch.empty((B, G, 3), dtype=torch.float32, device=xyz.device)
    call("mla_lga_prep", _p(xyz), _p(feats_bnc), _p(fps_idx), _p(knn_idx), _p(rows), _p(lc), B, N, G, K, C, float(alpha),
         float(beta))
    return rows, lc


def colstats(x2d):
    rows, C = x2d.shape
    mean = torch.empty(C, dtype=torch.float32, device=x2d.device)
    var = torch.empty(C, dtype=torch.float32, device=x2d.device)
    P = lib().mla_colstats_blocks(rows)
    ws = workspace(P * 2 * C * 4, x2d.device)
    call("mla_colstats_bf16", _p(x2d), _p(mean), _p(var), rows, C, x2d.stride(0), _p(ws), ws.numel())
    return mean, var


def bn_apply(x2d, mean, var, w, b, eps, residual=None, relu=False):
    rows, C = x2d.shape
    y = torch.empty_like(x2d)
    call("mla_bn_apply", _p(x2d), _p(mean), _p(var), _p(w), _p(b), _p(residual), _p(y), rows, C, float(eps), 1 if relu else 0)
    return y


def maxpool_k(x2d, groups, K):
    C = x2d.shape[1]
    out = torch.empty((groups, C), dtype=torch.bfloat16, device=x2d.device)
    call("mla_maxpool_k", _p(x2d), _p(out), groups, K, C)
    return out


def im2col_patch(pix, P, Kpad):
    B, CT, Himg, Wimg = pix.shape
    rows = torch.empty((B * (Himg // P) * (Wimg // P), Kpad), dtype=torch.bfloat16, device=pix.device)
    call("mla_im2col_patch", _p(pix), 1 if pix.dtype == torch.float32 else 0, _p(rows), B, CT, Himg, Wimg, P, Kpad)
    return rows


def avgpool_tokens(x, B, gh, gw, cs):
    C = x.shape[-1]
    y = torch.empty((B * (gh // cs) * (gw // cs), C), dtype=torch.bfloat16, device=x.device)
    call("mla_avgpool_tokens", _p(x), _p(y), B, gh, gw, C, cs)
    return y


def local_attn(q, kv, B, gh, gw, cs, heads, scale):
    C = q.shape[-1]
    out = torch.empty_like(q)
    call("mla_local_attn", _p(q), _p(kv), _p(out), B, gh, gw, C, cs, heads, float(scale))
    return out


# --------------------------------------------------------------------------------------------- generation heads
from ctypes import c_ulonglong  # noqa: E402

register_signatures({
    "mla_gemm_batched_bf16": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                              c_int, c_int, c_longlong, c_longlong, c_longlong, c_longlong, c_longlong, c_longlong, c_void_p],
    "mla_softmax_rows_fwd": [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_float, c_ulonglong, c_void_p],
    "mla_softmax_rows_bwd": [c_void_p, c_int, c_void_p, c_void_p, c_longlong, c_int, c_int, c_float, c_ulonglong, c_void_p],
    "mla_dropout_fwd": [c_void_p, c_void_p, c_void_p, c_longlong, c_float, c_ulonglong, c_void_p],
    "mla_dropout_bwd": [c_void_p, c_void_p, c_longlong, c_float, c_ulonglong, c_void_p],
    "mla_scale_batch": [c_void_p, c_void_p, c_void_p, c_longlong, c_longlong, c_void_p],
    "mla_layernorm_stats_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_float, c_void_p],
    "mla_layernorm_bwd_blocks": [c_longlong],
    "mla_layernorm_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_int,
                          c_void_p, c_size_t, c_void_p],
    "mla_seqmean_fwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mla_seqmean_bwd": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mla_bn_bwd_blocks": [c_longlong],
    "mla_bn_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_int, c_float,
                   c_void_p, c_size_t, c_void_p],
    "mla_chamfer_fwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t,
                        c_void_p],
    "mla_chamfer_bwd": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mla_imgloss_fwd": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p,
                        c_size_t, c_void_p],
    "mla_imgroi_fwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                       c_int, c_int, c_float, c_float, c_void_p, c_size_t, c_void_p],
    "mla_imgroi_bwd": [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                       c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p],
    "mla_imgloss_bwd": [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                        c_void_p],
})


def gemm_batched(a, b, out, *, M, N, K, lda, ldb, ldc, a_mode=0, b_mode=0, alpha=1.0, n_outer, n_inner, sA, sB, sC):
    """out[o,i] = alpha * Aop[o,i] . Bop[o,i]^T over a two-level batch; a/b/out are base tensors (offset = data_ptr), strides in
    elements as (outer, inner) pairs. out may be bf16 or fp32."""
    _req(a, torch.bfloat16, "gemm_batched A")
    _req(b, torch.bfloat16, "gemm_batched B")
    out_fp32 = 1 if out.dtype == torch.float32 else 0
    if not out_fp32:
        _req(out, torch.bfloat16, "gemm_batched C")
    call("mla_gemm_batched_bf16", _p(a), _p(b), _p(out), M, N, K, lda, ldb, ldc, a_mode, b_mode, out_fp32, float(alpha), n_outer,
         n_inner, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1])
    return out


def softmax_rows_fwd(scores, nvalid, p, seed):
    ncols = scores.shape[-1]
    rows = scores.numel() // ncols
    P = torch.empty(scores.shape, dtype=torch.bfloat16, device=scores.device)
    Pd = torch.empty_like(P) if p > 0 else P
    call("mla_softmax_rows_fwd", _p(scores), _p(P), _p(Pd), rows, ncols, nvalid, float(p), seed)
    return P, Pd


def softmax_rows_bwd(dPd, P, nvalid, p, seed):
    ncols = P.shape[-1]
    dS = torch.empty_like(P)
    call("mla_softmax_rows_bwd", _p(dPd), 1 if dPd.dtype == torch.float32 else 0, _p(P), _p(dS), P.numel() // ncols, ncols, nvalid, float(p), seed)
    return dS


def dropout_fwd(x, residual, p, seed):
    y = torch.empty_like(x)
    call("mla_dropout_fwd", _p(x), _p(residual), _p(y), x.numel(), float(p), seed)
    return y


def dropout_bwd(dy, p, seed):
    dx = torch.empty_like(dy)
    call("mla_dropout_bwd", _p(dy), _p(dx), dy.numel(), float(p), seed)
    return dx


def scale_batch(x, scale):
    y = torch.empty_like(x)
    call("mla_scale_batch", _p(x), _p(scale), _p(y), x.shape[0], x.numel() // x.shape[0])
    return y


def layernorm_stats_fwd(x2d, w, b, eps):
    rows, H = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty_like(mean)
    call("mla_layernorm_stats_fwd", _p(x2d), _p(w), _p(b), _p(y), _p(mean), _p(rstd), rows, H, float(eps))
    return y, mean, rstd


def layernorm_bwd(dy2d, x2d, w, mean, rstd, dw=None, db=None, accumulate=False):
    rows, H = x2d.shape
    dx = torch.empty_like(x2d)
    nb = lib().mla_layernorm_bwd_blocks(rows)
    ws = workspace(nb * 2 * H * 4, x2d.device)
    call("mla_layernorm_bwd", _p(dy2d), _p(x2d), _p(w), _p(mean), _p(rstd), _p(dx), _p(dw), _p(db), 1 if accumulate else 0, rows, H,
         _p(ws), ws.numel())
    return dx


def seqmean_fwd(x3d):
    B, S, C = x3d.shape
    y = torch.empty((B, C), dtype=torch.bfloat16, device=x3d.device)
    call("mla_seqmean_fwd", _p(x3d), _p(y), B, S, C)
    return y


def seqmean_bwd(dy2d, S):
    B, C = dy2d.shape
    dx = torch.empty((B, S, C), dtype=torch.bfloat16, device=dy2d.device)
    call("mla_seqmean_bwd", _p(dy2d), _p(dx), B, S, C)
    return dx


def bn_bwd(dy2d, x2d, mean, var, w, eps, dw=None, db=None, accumulate=False):
    rows, C = x2d.shape
    dx = torch.empty_like(x2d)
    nb = lib().mla_bn_bwd_blocks(rows)
    ws = workspace((nb * 2 * C + 2 * C) * 4, x2d.device)
    call("mla_bn_bwd", _p(dy2d), _p(x2d), _p(mean), _p(var), _p(w), _p(dx), _p(dw), _p(db), 1 if accumulate else 0, rows, C, float(eps),
         _p(ws), ws.numel())
    return dx


def chamfer_fwd(pred, gt):
    B, N, _ = pred.shape
    M = gt.shape[1]
    dev = pred.device
    d1 = torch.empty((B, N), dtype=torch.float32, device=dev)
    i1 = torch.empty((B, N), dtype=torch.int32, device=dev)
    d2 = torch.empty((B, M), dtype=torch.float32, device=dev)
    i2 = torch.empty((B, M), dtype=torch.int32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    ws = workspace(64, dev)
    call("mla_chamfer_fwd", _p(pred), _p(gt), _p(d1), _p(i1), _p(d2), _p(i2), _p(loss), B, N, M, _p(ws), ws.numel())
    return loss, d1, i1, d2, i2


def chamfer_bwd(pred, gt, d1, i1, d2, i2, gscale):
    B, N, _ = pred.shape
    dpred = torch.empty_like(pred)
    call("mla_chamfer_bwd", _p(pred), _p(gt), _p(d1), _p(i1), _p(d2), _p(i2), _p(gscale), _p(dpred), B, N, gt.shape[1])
    return dpred


def imgloss_fwd(delta_raw, curr, nxt, ps, clip):
    """delta_raw [B, npatch, ld] bf16 with ld >= 3*ps*ps (columns beyond are padding)."""
    B, ld = delta_raw.shape[0], delta_raw.shape[-1]
    fp32 = curr.dtype == torch.float32
    if curr.dtype != nxt.dtype or curr.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("imgloss: curr/next images must both be fp32 or both bf16")
    sums = torch.empty(3, dtype=torch.float32, device=delta_raw.device)
    ws = workspace(2048 * 3 * 4, delta_raw.device)
    call("mla_imgloss_fwd", _p(delta_raw), ld, _p(curr), _p(nxt), 1 if fp32 else 0, _p(sums), B, curr.shape[1], nxt.shape[1], curr.shape[2],
         ps, float(clip), _p(ws), ws.numel())
    return sums


def imgloss_bwd(delta_raw, curr, nxt, ps, clip, gscale):
    ld = delta_raw.shape[-1]
    out = torch.zeros_like(delta_raw) if ld != 3 * ps * ps else torch.empty_like(delta_raw)
    call("mla_imgloss_bwd", _p(delta_raw), ld, _p(curr), _p(nxt), 1 if curr.dtype == torch.float32 else 0, _p(gscale), _p(out),
         delta_raw.shape[0], curr.shape[1], nxt.shape[1], curr.shape[2], ps, float(clip))
    return out


def local_attn_bwd(q, kv, dout, B, gh, gw, cs, heads, scale):
    C = q.shape[1]
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    call("mla_local_attn_bwd", _p(q), _p(kv), _p(dout), _p(dq), _p(dkv), B, gh, gw, C, cs, heads, float(scale))
    return dq, dkv


def avgpool_tokens_bwd(dy, other, B, gh, gw, cs):
    C = dy.shape[1]
    dx = torch.empty((B * gh * gw, C), dtype=torch.bfloat16, device=dy.device)
    call("mla_avgpool_tokens_bwd", _p(dy), _p(other), _p(dx), B, gh, gw, C, cs)
    return dx


def imgroi_fwd(delta_raw, a_raw, o_raw, roi_u8, curr, nxt, ps, clip, shift):
    """sums[4] = ROI sum diff^2, ROI sum |diff|, background sum |diff|, sum |delta| (see mla_imgroi_fwd)."""
    B, npatch, ld = delta_raw.shape
    if curr.dtype != nxt.dtype or curr.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("imgroi: curr/next images must both be fp32 or both bf16")
    sums = torch.empty(4, dtype=torch.float32, device=delta_raw.device)
    ws = workspace(B * npatch * 16, delta_raw.device)
    call("mla_imgroi_fwd", _p(delta_raw), ld, _p(a_raw), a_raw.stride(0), _p(o_raw), o_raw.stride(0), _p(roi_u8), _p(curr), _p(nxt),
         1 if curr.dtype == torch.float32 else 0, _p(sums), B, curr.shape[1], nxt.shape[1], curr.shape[2], ps, float(clip), float(shift), _p(ws),
         ws.numel())
    return sums


def imgroi_bwd(delta_raw, a_raw, o_raw, roi_u8, curr, nxt, ps, clip, shift, coef):
    B, npatch, ld = delta_raw.shape
    dev = delta_raw.device
    dd = torch.empty_like(delta_raw)
    da = torch.empty(B * npatch, dtype=torch.float32, device=dev)
    do = torch.empty((B * npatch, 2), dtype=torch.float32, device=dev)
    call("mla_imgroi_bwd", _p(delta_raw), ld, _p(a_raw), a_raw.stride(0), _p(o_raw), o_raw.stride(0), _p(roi_u8), _p(curr), _p(nxt),
         1 if curr.dtype == torch.float32 else 0, _p(coef), _p(dd), _p(da), _p(do), B, curr.shape[1], nxt.shape[1], curr.shape[2], ps,
         float(clip), float(shift))
    return dd, da, do


def clip_preprocess(img_u8, bounds_h, coef_h, bounds_v, coef_v, OH, OW, mean, std, out_dtype=torch.float32, mask_channel=True):
    """img_u8 [B, H, W, 3] uint8 on the GPU -> [B, 3 (+1), OH, OW]; tables from mla_amd.vision_tokenizer.pil_resample_tables."""
    if img_u8.dtype != torch.uint8 or not img_u8.is_cuda:
        raise TypeError("clip_preprocess: uint8 CUDA tensor [B, H, W, 3] expected")
    B, H, W, _ = img_u8.shape
    out = torch.empty((B, 4 if mask_channel else 3, OH, OW), dtype=out_dtype, device=img_u8.device)
    m = (c_float * 3)(*[float(v) for v in mean])
    sd = (c_float * 3)(*[float(v) for v in std])
    call("mla_clip_preprocess", _p(img_u8.contiguous()), B, H, W, _p(bounds_h), _p(coef_h), coef_h.shape[1], _p(bounds_v), _p(coef_v),
         coef_v.shape[1], _p(out), 1 if out_dtype == torch.float32 else 0, OH, OW, m, sd, 1 if mask_channel else 0)
    return out


def lga_prep_bwd(drows, fps_idx, knn_idx, B, N, C):
    G, K = knn_idx.shape[1], knn_idx.shape[2]
    dfeats = torch.empty((B, N, C), dtype=torch.float32, device=drows.device)    # every element is written (gather form)
    call("mla_lga_prep_bwd", _p(drows), _p(fps_idx), _p(knn_idx), _p(dfeats), B, N, G, K, C)
    return dfeats


def maxpool_k_bwd(x2d, dy, groups, K):
    dx = torch.empty_like(x2d)
    call("mla_maxpool_k_bwd", _p(x2d), _p(dy), _p(dx), groups, K, x2d.shape[1])
    return dx
