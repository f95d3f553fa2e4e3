// Shared device/host helpers for libmla_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define MLA_GLOBAL_AS __attribute__((address_space(1)))
#define MLA_LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// float -> bfloat16, round-to-nearest-even (matches torch's cast): gfx950 has the conversion in hardware
// (v_cvt_pk_bf16_f32, two values per instruction) -- the integer emulation costs ~6 VALU instructions per value, which made the
// attention kernels VALU-bound (687 VALU per 64 MFMAs in the forward loop).
typedef __attribute__((ext_vector_type(2))) float mla_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 mla_bf16x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const mla_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, mla_bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, f) & 0xffffu); }
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// 64-lane wave reductions (xor butterflies; every lane ends with the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 32 lanes of this lane's half of the wave (lanes 0..31 / 32..63) on the VALU's DPP path (no LDS crossbar round trips:
// five dependent ds_bpermute per row cost the GEMM epilogue ~4 us per tile). The total is valid in the LAST lane of the half only
// (lane 31 / 63); fixed order -> deterministic. row_shr:n = 0x110 + n, row_bcast:15 = 0x142 (rows 1 and 3 take lane 15 of rows 0 and 2).
__device__ __forceinline__ float half_wave_sum_last(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, true));
#endif
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block reduction through LDS scratch (>= 16 floats). All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = scratch[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
  return r;
}

// Work distribution of the streaming kernels: workgroup b owns the contiguous item range [b * chunk, (b + 1) * chunk) instead of
// the usual grid-stride walk. With a grid-stride loop all CUs sweep one narrow window through every stream in lockstep and the
// rate then depends on how the streams' physical addresses happen to relate (measured on AdamW's five streams: 4.5-4.6 TB/s vs
// 5.7-6.2 TB/s on the same buffers, tools/exp_skew.py); contiguous chunks keep the active workgroups spread over all of memory.
// Used by AdamW only: SwiGLU / RoPE (two or three bf16 streams with a row structure) measured 4-7 % SLOWER with it in the step.
#define MLA_CHUNK_LOOP(IDX, TOTAL)                                                                            \
  const long long chunk__ = ((((TOTAL) + gridDim.x - 1) / gridDim.x) + 255) & ~255LL;                         \
  const long long lo__ = (long long)blockIdx.x * chunk__, hi__ = lo__ + chunk__ < (TOTAL) ? lo__ + chunk__ : (TOTAL); \
  for (long long IDX = lo__ + threadIdx.x; IDX < hi__; IDX += 256)

// 8 bf16 <-> 8 floats (one 16-B access)
__device__ __forceinline__ void unpack8(const u32x4_t& w, float* f) {
  f[0] = bflo(w[0]); f[1] = bfhi(w[0]); f[2] = bflo(w[1]); f[3] = bfhi(w[1]);
  f[4] = bflo(w[2]); f[5] = bfhi(w[2]); f[6] = bflo(w[3]); f[7] = bfhi(w[3]);
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
  u32x4_t w;
  w[0] = pack2bf(f[0], f[1]); w[1] = pack2bf(f[2], f[3]); w[2] = pack2bf(f[4], f[5]); w[3] = pack2bf(f[6], f[7]);
  return w;
}

// SwiGLU forward of one element: silu(g) * u on bf16-rounded inputs, fp32 arithmetic, rounded once by the caller
// (LlamaMLP.forward, modeling_llama.py:240). One definition for the stand-alone kernels and the fused GEMM epilogue.
// sigmoid: v_exp_f32 + v_rcp_f32 (1 ulp). The IEEE-exact `1.f / x` compiles to v_div_scale x 2 + v_rcp + 4 fma + v_div_fmas +
// v_div_fixup = 9 more VALU per element -- in the fused GEMM epilogues that was a third of the arithmetic the CU does while its
// matrix pipe idles (round 3). The result is rounded to bf16 right after; fused and stand-alone kernels share this definition.
__device__ __forceinline__ float mla_sigmoid(float g) { return __builtin_amdgcn_rcpf(1.f + __expf(-g)); }
__device__ __forceinline__ float swiglu_fwd_elem(float g, float u) {
  const float sg = mla_sigmoid(g);
  return (g * sg) * u;
}

// SwiGLU backward of one element (autograd of LlamaMLP.forward, modeling_llama.py:240): d = d(act), act = silu(g) * u.
// ONE definition for the stand-alone kernel (transpose.hip) and the fused GEMM epilogue (gemm256.hip): identical results, bit for bit.
__device__ __forceinline__ void swiglu_bwd_elem(float d, float g, float u, float& dg, float& du) {
  const float sg = mla_sigmoid(g);
  const float sl = g * sg;
  dg = d * u * (sg + sl * (1.f - sg));
  du = d * sl;
}

// async global -> LDS copy of 16 B per lane: LDS destination = wave-uniform base + lane*16.
#ifndef MLA_GLDS_AUX
#define MLA_GLDS_AUX 0   // cache-policy bits of the LDS-DMA loads. Measured on gemm256 (4 shapes, A/B in one box): 1 (sc0) +-0.5 %,
                         // 2 (nt) and 3 (sc0 + nt) -5 ... -25 % -- the operand panels are re-read by 4-8 CUs through the XCD's L2
#endif
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const MLA_GLOBAL_AS void*)gsrc, (MLA_LDS_AS void*)lds_wave_base, 16, 0, MLA_GLDS_AUX);
}

// The same copy issued from inline assembly, i.e. invisible to the compiler's wait-count insertion. With the builtin the compiler
// cannot tell which LDS bytes an LDS-DMA load writes, so it puts `s_waitcnt vmcnt(0)` in front of the first LDS read it cannot
// disambiguate (every ds_read_b64_tr_b16, which carries no memory operand): in the attention loops that drained the next tile's
// prefetch in the middle of the current tile. Callers order the copy against its readers themselves (vmcnt wait + barrier).
//
// M0 CONTRACT: the statement overwrites m0 and cannot say so -- hipcc treats m0 as reserved ("m0" in the clobber list only draws
// `-Winline-asm: clobber list contains reserved registers`, and the note says the clobber may be ignored). Compiled code uses m0
// for exactly one thing on gfx950 in this library: the LDS base of the glds16() BUILTIN, which it may keep live across statements.
// Therefore A KERNEL USES EITHER glds16 OR glds16_untracked, NEVER BOTH: every call site selects the variant by a template
// parameter / macro that is fixed per kernel (attention.hip ATTN_GLDS, gemm.hip UNTRACKED, gemm256.hip UNTRACKED), and m0 is
// written in the SAME statement that reads it, so nothing depends on its value before or after.
__device__ __forceinline__ void glds16_untracked(const void* gsrc, void* lds_wave_base) {
  const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(MLA_LDS_AS void*)lds_wave_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m0v) : "memory");
}

// LDS transpose read: 4 x b16 per lane (see DESIGN.md "tr16 semantics", verified by mla_selftest_tr16)
__device__ __forceinline__ short4_t lds_tr16_b64(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((MLA_LDS_AS short4_t*)lds_addr);
}

// ---- host side error plumbing (see include/mla_hip.h) ----
void mla_set_error(const char* fmt, ...);
#define MLA_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      mla_set_error(__VA_ARGS__);                \
      return -1;                                 \
    }                                            \
  } while (0)
#define MLA_LAUNCH_CHECK()                                            \
  do {                                                                \
    hipError_t e__ = hipGetLastError();                               \
    if (e__ != hipSuccess) {                                          \
      mla_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
      return (int)e__;                                                \
    }                                                                 \
    return 0;                                                         \
  } while (0)
