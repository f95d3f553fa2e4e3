// Tile transposes feeding the wgrad / dgrad GEMMs (DESIGN.md "all-NT backward").
// The MFMA fragment of an operand needs 8 consecutive reduction elements per lane. For the backward products
// (dx = dy W, dW = dy^T x) the natural layouts are reduction-major, which forces ds_read_b64_tr_b16 gathers; those
// reads are issue-limited at the 2 waves/SIMD a 256x256 tile leaves (measured: 470-700 TFLOP/s vs 1100-1250 with
// ds_read_b128 on the same bytes, 0 bank conflicts either way). So the backward GEMMs are fed k-contiguous operands
// instead: W^T, x^T and dy^T are produced by these HBM-bound kernels (64x64 tiles through LDS, 16-B global accesses
// on both sides), two of them fused with the elementwise op that (re)computes the tensor anyway.
#include "common.h"

namespace {

struct CopyOp {
  __device__ __forceinline__ void load8(const bf16_t* src, long long ld, long long r, int c, float* v) const {
    const u32x4_t w = *(const u32x4_t*)(src + r * ld + c);
    v[0] = bflo(w[0]); v[1] = bfhi(w[0]); v[2] = bflo(w[1]); v[3] = bfhi(w[1]);
    v[4] = bflo(w[2]); v[5] = bfhi(w[2]); v[6] = bflo(w[3]); v[7] = bfhi(w[3]);
  }
};
// y = w * bf16(x * rstd[row])  (LlamaRMSNorm with the statistic saved by the forward pass)
struct RmsApplyOp {
  const bf16_t* w; const float* rstd;
  __device__ __forceinline__ void load8(const bf16_t* src, long long ld, long long r, int c, float* v) const {
    const u32x4_t x = *(const u32x4_t*)(src + r * ld + c);
    const u32x4_t ww = *(const u32x4_t*)(w + c);
    const float rs = rstd[r];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = bflo(ww[j]) * bf2f(f2bf(bflo(x[j]) * rs));
      v[2 * j + 1] = bfhi(ww[j]) * bf2f(f2bf(bfhi(x[j]) * rs));
    }
  }
};
// act = silu(gate) * up from gu = [gate | up] (row stride ld = 2I); logical source width = I
struct SwigluOp {
  int I;
  __device__ __forceinline__ void load8(const bf16_t* src, long long ld, long long r, int c, float* v) const {
    const u32x4_t g = *(const u32x4_t*)(src + r * ld + c);
    const u32x4_t u = *(const u32x4_t*)(src + r * ld + I + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g0 = bflo(g[j]), g1 = bfhi(g[j]);
      v[2 * j] = swiglu_fwd_elem(g0, bflo(u[j]));
      v[2 * j + 1] = swiglu_fwd_elem(g1, bfhi(u[j]));
    }
  }
};

// dst[c][r] = op(src)[r][c];  src logical [R, C] (row stride ld), dst [C, R] (row stride ldt). R % 8 == 0, C % 8 == 0.
template <typename OP>
__global__ __launch_bounds__(256) void tile_transpose_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, long long R,
                                                             int C, long long ld, long long ldt, OP op) {
  __shared__ unsigned short t[64][66];
  const long long r0 = (long long)blockIdx.y * 64;
  const int c0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int rr = (tid >> 3) + it * 32, cc = (tid & 7) * 8;
    if (r0 + rr < R && c0 + cc < C) {
      float v[8];
      op.load8(src, ld, r0 + rr, c0 + cc, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) t[cc + j][rr] = f2bf(v[j]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int cc = (tid >> 3) + it * 32, rr = (tid & 7) * 8;
    if (c0 + cc < C && r0 + rr < R) {
      const uint32_t* p = (const uint32_t*)&t[cc][rr];
      u32x4_t w = {p[0], p[1], p[2], p[3]};
      *(u32x4_t*)(dst + (long long)(c0 + cc) * ldt + r0 + rr) = w;
    }
  }
}

// SwiGLU backward with both layouts in one pass (replaces swiglu_bwd + a 2-stream transpose of its 2I-wide output):
//   dgu[t][i] = dact * up * silu'(gate), dgu[t][I + i] = dact * silu(gate)   (row-major, feeds the dgrad GEMM)
//   dguT[i][t], dguT[I + i][t] = the same values, token-contiguous            (feeds the wgrad GEMM)
// 64 x 64 tiles over [rows, I]; the row-major halves are stored straight from registers, the transposed ones through LDS.
__global__ __launch_bounds__(256) void swiglu_bwd_t_kernel(const bf16_t* __restrict__ dact, const bf16_t* __restrict__ gu,
                                                           bf16_t* __restrict__ dgu, bf16_t* __restrict__ dguT, long long R, int I,
                                                           long long ldt) {
  __shared__ unsigned short tg[64][66], tu[64][66];
  const long long r0 = (long long)blockIdx.y * 64;
  const int c0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int rr = (tid >> 3) + it * 32, cc = (tid & 7) * 8;
    const long long r = r0 + rr;
    const int c = c0 + cc;
    if (r < R && c < I) {
      const u32x4_t g = *(const u32x4_t*)(gu + r * 2 * I + c);
      const u32x4_t u = *(const u32x4_t*)(gu + r * 2 * I + I + c);
      const u32x4_t d = *(const u32x4_t*)(dact + r * I + c);
      float gv[8], uv[8], dv[8], dg[8], du[8];
      unpack8(g, gv); unpack8(u, uv); unpack8(d, dv);
#pragma unroll
      for (int j = 0; j < 8; ++j) swiglu_bwd_elem(dv[j], gv[j], uv[j], dg[j], du[j]);
      const u32x4_t pg = pack8(dg), pu = pack8(du);
      *(u32x4_t*)(dgu + r * 2 * I + c) = pg;
      *(u32x4_t*)(dgu + r * 2 * I + I + c) = pu;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        tg[cc + 2 * j][rr] = (unsigned short)(pg[j] & 0xffffu); tg[cc + 2 * j + 1][rr] = (unsigned short)(pg[j] >> 16);
        tu[cc + 2 * j][rr] = (unsigned short)(pu[j] & 0xffffu); tu[cc + 2 * j + 1][rr] = (unsigned short)(pu[j] >> 16);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int cc = (tid >> 3) + it * 32, rr = (tid & 7) * 8;
    if (c0 + cc < I && r0 + rr < R) {
      const uint32_t* pg = (const uint32_t*)&tg[cc][rr];
      const uint32_t* pu = (const uint32_t*)&tu[cc][rr];
      *(u32x4_t*)(dguT + (long long)(c0 + cc) * ldt + r0 + rr) = u32x4_t{pg[0], pg[1], pg[2], pg[3]};
      *(u32x4_t*)(dguT + (long long)(I + c0 + cc) * ldt + r0 + rr) = u32x4_t{pu[0], pu[1], pu[2], pu[3]};
    }
  }
}

// SwiGLU forward writing BOTH layouts in one pass: act[t][i] (row-major, feeds the down-projection GEMM) and actT[i][t]
// (token-contiguous, the wgrad operand of the backward). The backward used to recompute the product from gu straight into the
// transposed layout (swiglu_fwd_t: read 2 I-wide streams, write 1): forward + recompute = 6 I-wide passes per layer, this = 4.
__global__ __launch_bounds__(256) void swiglu_fwd_dual_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act,
                                                              bf16_t* __restrict__ actT, long long R, int I, long long ldt) {
  __shared__ unsigned short t[64][66];
  const long long r0 = (long long)blockIdx.y * 64;
  const int c0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int rr = (tid >> 3) + it * 32, cc = (tid & 7) * 8;
    const long long r = r0 + rr;
    const int c = c0 + cc;
    if (r < R && c < I) {
      float g[8], u[8], o[8];
      unpack8(*(const u32x4_t*)(gu + r * 2 * I + c), g);
      unpack8(*(const u32x4_t*)(gu + r * 2 * I + I + c), u);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = swiglu_fwd_elem(g[j], u[j]);
      const u32x4_t pk = pack8(o);
      *(u32x4_t*)(act + r * I + c) = pk;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[cc + 2 * j][rr] = (unsigned short)(pk[j] & 0xffffu);
        t[cc + 2 * j + 1][rr] = (unsigned short)(pk[j] >> 16);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int cc = (tid >> 3) + it * 32, rr = (tid & 7) * 8;
    if (c0 + cc < I && r0 + rr < R) {
      const uint32_t* q = (const uint32_t*)&t[cc][rr];
      *(u32x4_t*)(actT + (long long)(c0 + cc) * ldt + r0 + rr) = u32x4_t{q[0], q[1], q[2], q[3]};
    }
  }
}

template <typename OP>
int launch_tt(const void* src, void* dst, long long R, int C, long long ld, long long ldt, OP op, hipStream_t stream, const char* who) {
  if (!(src && dst && R > 0 && C > 0 && R % 8 == 0 && C % 8 == 0 && ld % 8 == 0 && ldt % 8 == 0)) {
    mla_set_error("%s: need R, C, ld, ldt multiples of 8", who);
    return -1;
  }
  if ((((uintptr_t)src) & 15) || (((uintptr_t)dst) & 15)) { mla_set_error("%s: 16-B alignment", who); return -1; }
  dim3 grid((C + 63) / 64, (unsigned)((R + 63) / 64));
  hipLaunchKernelGGL(tile_transpose_kernel<OP>, grid, dim3(256), 0, stream, (const bf16_t*)src, (bf16_t*)dst, R, C, ld, ldt, op);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mla_set_error("%s: launch failed: %s", who, hipGetErrorString(e)); return (int)e; }
  return 0;
}

}  // namespace

extern "C" int mla_transpose_bf16(const void* src, void* dst, long long R, int C, long long ld, long long ldt, hipStream_t stream) {
  return launch_tt(src, dst, R, C, ld, ldt, CopyOp{}, stream, "mla_transpose_bf16");
}
// dst[h][t] = w[h] * bf16(x[t][h] * rstd[t])
extern "C" int mla_rmsnorm_apply_t(const void* x, const void* w, const float* rstd, void* dst, long long rows, int H, long long ldt,
                                   hipStream_t stream) {
  if (!w || !rstd) { mla_set_error("mla_rmsnorm_apply_t: null pointer"); return -1; }
  return launch_tt(x, dst, rows, H, H, ldt, RmsApplyOp{(const bf16_t*)w, rstd}, stream, "mla_rmsnorm_apply_t");
}
// dst[i][t] = silu(gu[t][i]) * gu[t][I + i]
extern "C" int mla_swiglu_fwd_t(const void* gu, void* dst, long long rows, int I, long long ldt, hipStream_t stream) {
  return launch_tt(gu, dst, rows, I, 2LL * I, ldt, SwigluOp{I}, stream, "mla_swiglu_fwd_t");
}

// act = silu(gate) * up [rows, I] and actT = its transpose [I, ldt >= rows], in one pass
extern "C" int mla_swiglu_fwd_dual(const void* gu, void* act, void* actT, long long rows, int I, long long ldt, hipStream_t stream) {
  if (!(gu && act && actT && rows > 0 && I > 0 && rows % 8 == 0 && I % 8 == 0 && ldt % 8 == 0 && ldt >= rows)) {
    mla_set_error("mla_swiglu_fwd_dual: need rows, I, ldt multiples of 8 and ldt >= rows");
    return -1;
  }
  if ((((uintptr_t)gu | (uintptr_t)act | (uintptr_t)actT) & 15) != 0) { mla_set_error("mla_swiglu_fwd_dual: 16-B alignment"); return -1; }
  dim3 grid((I + 63) / 64, (unsigned)((rows + 63) / 64));
  hipLaunchKernelGGL(swiglu_fwd_dual_kernel, grid, dim3(256), 0, stream, (const bf16_t*)gu, (bf16_t*)act, (bf16_t*)actT, rows, I, ldt);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mla_set_error("mla_swiglu_fwd_dual: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}

// dgu = SwiGLU backward of dact (row-major [rows, 2I]) and dguT = its transpose [2I, ldt >= rows], in one pass
extern "C" int mla_swiglu_bwd_t(const void* dact, const void* gu, void* dgu, void* dguT, long long rows, int I, long long ldt,
                                hipStream_t stream) {
  if (!(dact && gu && dgu && dguT && rows > 0 && I > 0 && rows % 8 == 0 && I % 8 == 0 && ldt % 8 == 0 && ldt >= rows)) {
    mla_set_error("mla_swiglu_bwd_t: need rows, I, ldt multiples of 8 and ldt >= rows");
    return -1;
  }
  if ((((uintptr_t)dact | (uintptr_t)gu | (uintptr_t)dgu | (uintptr_t)dguT) & 15) != 0) { mla_set_error("mla_swiglu_bwd_t: 16-B alignment"); return -1; }
  dim3 grid((I + 63) / 64, (unsigned)((rows + 63) / 64));
  hipLaunchKernelGGL(swiglu_bwd_t_kernel, grid, dim3(256), 0, stream, (const bf16_t*)dact, (const bf16_t*)gu, (bf16_t*)dgu, (bf16_t*)dguT,
                     rows, I, ldt);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { mla_set_error("mla_swiglu_bwd_t: launch failed: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}
