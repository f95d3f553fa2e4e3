// HBM-bound kernels of the decoder / heads: every kernel moves 16 B per lane per access (8 x bf16),
// accumulates in fp32 and makes exactly one pass over its inputs (DESIGN.md "HBM-bound kernels").
#include "common.h"
#include <math.h>

namespace {


// ---------------------------------------------------------------------------------------------------------
// RMSNorm  (reference: transformers/models/llama/modeling_llama.py:76-90; timm RmsNorm used by FinalLayer,
// models/diffusion/models.py:177).  y = w * bf16(x * rsqrt(mean(x^2) + eps))  -- cast happens BEFORE the weight
// multiply.  One 256-thread block per row, H <= 8192, H % 8 == 0.
// ---------------------------------------------------------------------------------------------------------
constexpr int NORM_MAXC = 4;

__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd_out,
                                                          int rows, int H, float eps) {
  __shared__ float scratch[16];
  const int row = blockIdx.x;
  const int nchunk = H >> 3;
  const bf16_t* xr = x + (size_t)row * H;
  float xv[NORM_MAXC][8];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NORM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      const u32x4_t v = *(const u32x4_t*)(xr + ch * 8);
      unpack8(v, xv[c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += xv[c][j] * xv[c][j];
    }
  }
  ss = block_sum(ss, scratch);
  const float rstd = 1.0f / sqrtf(ss / (float)H + eps);
  if (threadIdx.x == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
  for (int c = 0; c < NORM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      float wv[8], o[8];
      unpack8(*(const u32x4_t*)(w + ch * 8), wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = wv[j] * bf2f(f2bf(xv[c][j] * rstd));
      *(u32x4_t*)(y + (size_t)row * H + ch * 8) = pack8(o);
    }
  }
}

// The two halves of a FOLDED RMSNorm (gemm_args.h nrm_* / rs_*, gemm256.hip mla_gemm_res_norm) for rows that do not come out of a GEMM
// epilogue (the first decoder layer's input; recomputation of a checkpointed layer): xg = bf16(x * w) -- the column scale, which does not
// commute with the projection -- and rstd per row, which the projection applies to its fp32 accumulator rows. One 256-thread block per row.
__global__ __launch_bounds__(256) void rmsnorm_prep_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                           bf16_t* __restrict__ xg, float* __restrict__ rstd_out, int rows, int H,
                                                           float eps) {
  __shared__ float scratch[16];
  const int row = blockIdx.x;
  const int nchunk = H >> 3;
  const bf16_t* xr = x + (size_t)row * H;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NORM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      float xv[8], wv[8], o[8];
      unpack8(*(const u32x4_t*)(xr + ch * 8), xv);
      unpack8(*(const u32x4_t*)(w + ch * 8), wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) { ss += xv[j] * xv[j]; o[j] = xv[j] * wv[j]; }
      *(u32x4_t*)(xg + (size_t)row * H + ch * 8) = pack8(o);
    }
  }
  ss = block_sum(ss, scratch);
  if (threadIdx.x == 0 && rstd_out) rstd_out[row] = 1.0f / sqrtf(ss / (float)H + eps);
}

// ---------------------------------------------------------------------------------------------------------
// timm==0.9.10 RmsNorm, the norm_final of the diffusion FinalLayer (models/diffusion/models.py:18,177; pin pyproject.toml:44).
// timm tag v0.9.10, timm/layers/norm.py::RmsNorm.forward -> timm/layers/fast_norm.py::fast_rms_norm -> (no apex) rms_norm:
//     v = torch.var(x, dim=-1, keepdim=True)      # UNBIASED and mean-subtracted: sum((x - mean)^2) / (H - 1)
//     x = x * torch.rsqrt(v + eps);  x = x * weight
// i.e. NOT the mean-of-squares RMS norm of LlamaRMSNorm (timm >= 1.0.13 fixed this and kept the 0.9 behaviour as "SimpleNorm").
// x itself is not centred. Rounding points follow the reference's bf16 autocast run: var and var + eps are bf16 tensors,
// torch.rsqrt is on autocast's fp32 list, so x * rsqrt(.) * weight is an fp32 product, cast to bf16 by fc1 (autocast Linear).
// One 256-thread block per row.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void timm_rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                               bf16_t* __restrict__ y, float* __restrict__ mean_out,
                                                               float* __restrict__ rstd_out, int rows, int H, float eps) {
  __shared__ float scratch[16];
  const int row = blockIdx.x;
  const int nchunk = H >> 3;
  const bf16_t* xr = x + (size_t)row * H;
  float xv[NORM_MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NORM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      unpack8(*(const u32x4_t*)(xr + ch * 8), xv[c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += xv[c][j];
    }
  }
  const float mean = block_sum(s, scratch) / (float)H;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NORM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = xv[c][j] - mean; ss += d * d; }
    }
  }
  const float var = bf2f(f2bf(block_sum(ss, scratch) / (float)(H - 1)));
  const float rstd = 1.0f / sqrtf(bf2f(f2bf(var + eps)));
  if (threadIdx.x == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
#pragma unroll
  for (int c = 0; c < NORM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      float wv[8], o[8];
      unpack8(*(const u32x4_t*)(w + ch * 8), wv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = xv[c][j] * rstd * wv[j];
      *(u32x4_t*)(y + (size_t)row * H + ch * 8) = pack8(o);
    }
  }
}

// y_j = x_j r w_j, r = (var + eps)^-1/2, var = sum_i (x_i - mean)^2 / (H - 1):
// dx_i = dy_i w_i r - r^3 (x_i - mean) / (H - 1) * sum_j dy_j w_j x_j ;  dw partial[blockIdx][j] = sum_rows dy_j x_j r
__global__ __launch_bounds__(256) void timm_rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                               const bf16_t* __restrict__ w, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                               float* __restrict__ dw_partial, int rows, int H) {
  __shared__ float scratch[16];
  const int nchunk = H >> 3;
  float wv[NORM_MAXC][8], dwacc[NORM_MAXC][8];
#pragma unroll
  for (int c = 0; c < NORM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
#pragma unroll
    for (int j = 0; j < 8; ++j) dwacc[c][j] = 0.f;
    if (ch < nchunk) unpack8(*(const u32x4_t*)(w + ch * 8), wv[c]);
  }
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const float rs = rstd[row], mu = mean[row];
    float xv[NORM_MAXC][8], dn[NORM_MAXC][8];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NORM_MAXC; ++c) {
      const int ch = threadIdx.x + c * 256;
      if (ch < nchunk) {
        float dyv[8];
        unpack8(*(const u32x4_t*)(x + (size_t)row * H + ch * 8), xv[c]);
        unpack8(*(const u32x4_t*)(dy + (size_t)row * H + ch * 8), dyv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          dn[c][j] = dyv[j] * wv[c][j];
          dot += dn[c][j] * xv[c][j];
          dwacc[c][j] += dyv[j] * xv[c][j] * rs;
        }
      }
    }
    dot = block_sum(dot, scratch) * rs * rs * rs / (float)(H - 1);
#pragma unroll
    for (int c = 0; c < NORM_MAXC; ++c) {
      const int ch = threadIdx.x + c * 256;
      if (ch < nchunk) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs * dn[c][j] - (xv[c][j] - mu) * dot;
        *(u32x4_t*)(dx + (size_t)row * H + ch * 8) = pack8(o);
      }
    }
  }
  if (dw_partial) {
#pragma unroll
    for (int c = 0; c < NORM_MAXC; ++c) {
      const int ch = threadIdx.x + c * 256;
      if (ch < nchunk) {
        float* o = dw_partial + (size_t)blockIdx.x * H + ch * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = dwacc[c][j];
      }
    }
  }
}

// dx = dres + rstd * (dy*w - n * mean(dy*w*n)),  n = x*rstd ;  dw partial[blockIdx][h] = sum_rows dy*n
// Each workgroup walks its rows with the NEXT row's three streams (x, dy, dres) already in flight while the current row is reduced
// and written: one row at a time left the loop waiting on HBM latency twice per row (load, then the block reduction's barrier).
template <int MAXC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ w, const float* __restrict__ rstd,
                                                          const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx,
                                                          float* __restrict__ dw_partial, int rows, int H) {
  __shared__ float scratch[16];
  const int nchunk = H >> 3;
  float wv[MAXC][8], dwacc[MAXC][8];
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
#pragma unroll
    for (int j = 0; j < 8; ++j) dwacc[c][j] = 0.f;
    if (ch < nchunk) unpack8(*(const u32x4_t*)(w + ch * 8), wv[c]);
  }
  u32x4_t nx[MAXC], ndy[MAXC], nres[MAXC];     // raw 16-B pieces of the row being prefetched
  auto fetch = [&](int row) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = threadIdx.x + c * 256;
      if (ch < nchunk && row < rows) {
        nx[c] = *(const u32x4_t*)(x + (size_t)row * H + ch * 8);
        ndy[c] = *(const u32x4_t*)(dy + (size_t)row * H + ch * 8);
        if (dres) nres[c] = *(const u32x4_t*)(dres + (size_t)row * H + ch * 8);
      }
    }
  };
  fetch(blockIdx.x);
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const float rs = rstd[row];
    float nv[MAXC][8], dn[MAXC][8], rv[MAXC][8];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = threadIdx.x + c * 256;
      if (ch < nchunk) {
        float xv[8], dyv[8];
        unpack8(nx[c], xv);
        unpack8(ndy[c], dyv);
        if (dres) unpack8(nres[c], rv[c]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          nv[c][j] = xv[j] * rs;
          dn[c][j] = dyv[j] * wv[c][j];
          dot += dn[c][j] * nv[c][j];
          dwacc[c][j] += dyv[j] * nv[c][j];
        }
      }
    }
    fetch(row + gridDim.x);                     // next row's loads fly during the reduction + stores of this one
    dot = block_sum(dot, scratch) / (float)H;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = threadIdx.x + c * 256;
      if (ch < nchunk) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (dres ? rv[c][j] : 0.f) + rs * (dn[c][j] - nv[c][j] * dot);
        *(u32x4_t*)(dx + (size_t)row * H + ch * 8) = pack8(o);
      }
    }
  }
  if (dw_partial) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int ch = threadIdx.x + c * 256;
      if (ch < nchunk) {
        float* o = dw_partial + (size_t)blockIdx.x * H + ch * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = dwacc[c][j];
      }
    }
  }
}

// out[n] (+)= sum_p partial[p][n]   (deterministic second stage for dw / bias-grad reductions)
// 16 columns x 16 row-lanes per block and four independent loads in flight per thread: with P up to 512 partial rows the old
// 64-column x 4-lane shape was a chain of 128 dependent-latency loads per thread on only N / 64 workgroups (27 us for N = 4096).
constexpr int RP_COLS = 16, RP_LANES = 16;
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                              int P, int N, int accumulate) {
  __shared__ float red[RP_LANES][RP_COLS + 1];
  const int c = threadIdx.x & (RP_COLS - 1), rl = threadIdx.x / RP_COLS;
  const int n = blockIdx.x * RP_COLS + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (n < N) {
    int p = rl;
    for (; p + 3 * RP_LANES < P; p += 4 * RP_LANES) {
      s0 += partial[(size_t)p * N + n];
      s1 += partial[(size_t)(p + RP_LANES) * N + n];
      s2 += partial[(size_t)(p + 2 * RP_LANES) * N + n];
      s3 += partial[(size_t)(p + 3 * RP_LANES) * N + n];
    }
    for (; p < P; p += RP_LANES) s0 += partial[(size_t)p * N + n];
  }
  red[rl][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < RP_LANES; ++i) t += red[i][c];      // fixed order: deterministic
    out[n] = accumulate ? out[n] + t : t;
  }
}

// partial[blockIdx.y][n] = sum over this block's row slice of dy[r][n]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ dy, float* __restrict__ partial,
                                                             int rows, int N, int ld) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const int rs = gridDim.y;
  const int per = (rows + rs - 1) / rs;
  const int r0 = blockIdx.y * per, r1 = (r0 + per) < rows ? (r0 + per) : rows;
  float s = 0.f;
  if (n < N)
    for (int r = r0 + rl; r < r1; r += 4) s += bf2f(dy[(size_t)r * ld + n]);
  red[rl][c] = s;
  __syncthreads();
  if (rl == 0 && n < N) partial[(size_t)blockIdx.y * N + n] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// same, 8 columns (16 B) per lane: 8 lanes cover the block's 64 columns, 32 row lanes stride over the row slice
__global__ __launch_bounds__(256) void colsum_partial_vec_kernel(const bf16_t* __restrict__ dy, float* __restrict__ partial,
                                                                 int rows, int N, int ld) {
  __shared__ float red[32][65];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int n = blockIdx.x * 64 + cl * 8;
  const int rs = gridDim.y;
  const int per = (rows + rs - 1) / rs;
  const int r0 = blockIdx.y * per, r1 = (r0 + per) < rows ? (r0 + per) : rows;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  if (n < N)
    for (int r = r0 + rl; r < r1; r += 32) {
      float v[8];
      unpack8(*(const u32x4_t*)(dy + (size_t)r * ld + n), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += v[j];
    }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[rl][cl * 8 + j] = s[j];
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < N) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red[i][threadIdx.x];
    partial[(size_t)blockIdx.y * N + blockIdx.x * 64 + threadIdx.x] = t;
  }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm forward (frozen vision tokenizer: models/mla/image/vision_tokenizer.py:21-24)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                            int rows, int H, float eps) {
  __shared__ float scratch[16];
  const int row = blockIdx.x;
  const int nchunk = H >> 3;
  float xv[NORM_MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NORM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      unpack8(*(const u32x4_t*)(x + (size_t)row * H + ch * 8), xv[c]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += xv[c][j];
    }
  }
  const float mean = block_sum(s, scratch) / (float)H;
  float vs = 0.f;
#pragma unroll
  for (int c = 0; c < NORM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = xv[c][j] - mean; vs += d * d; }
    }
  }
  const float var = block_sum(vs, scratch) / (float)H;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int c = 0; c < NORM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      float wv[8], bv[8], o[8];
      unpack8(*(const u32x4_t*)(w + ch * 8), wv);
      unpack8(*(const u32x4_t*)(b + ch * 8), bv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (xv[c][j] - mean) * rstd * wv[j] + bv[j];
      *(u32x4_t*)(y + (size_t)row * H + ch * 8) = pack8(o);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// RoPE, half-split rotation, applied in place to the q and k slices of the packed qkv buffer
// (reference: modeling_llama.py:96-145, 177-208; positions = arange(S) for every row, :985-990).
// sign = +1 forward, -1 backward (the transpose of a rotation).  tab = [S][D/2] fp32 cos and sin.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ buf, const float* __restrict__ cos_t,
                                                   const float* __restrict__ sin_t, long long tokens, int S, int nheads,
                                                   int D, int ld, int q_off, int k_off, float sign) {
  const int half = D >> 1, cpd = half >> 3;  // 16-B chunks per half head
  const long long total = tokens * 2 * nheads * cpd;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int dc = (int)(idx % cpd);
    long long r = idx / cpd;
    const int h = (int)(r % nheads);
    r /= nheads;
    const int which = (int)(r & 1);
    const long long t = r >> 1;
    const int pos = (int)(t % S);
    bf16_t* p = buf + t * ld + (which ? k_off : q_off) + h * D + dc * 8;
    float a[8], b[8], c[8], s[8], oa[8], ob[8];
    unpack8(*(const u32x4_t*)p, a);
    unpack8(*(const u32x4_t*)(p + half), b);
    const f32x4_t c0 = *(const f32x4_t*)(cos_t + (size_t)pos * half + dc * 8);
    const f32x4_t c1 = *(const f32x4_t*)(cos_t + (size_t)pos * half + dc * 8 + 4);
    const f32x4_t s0 = *(const f32x4_t*)(sin_t + (size_t)pos * half + dc * 8);
    const f32x4_t s1 = *(const f32x4_t*)(sin_t + (size_t)pos * half + dc * 8 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = c0[j]; c[j + 4] = c1[j]; s[j] = s0[j] * sign; s[j + 4] = s1[j] * sign; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      oa[j] = fmaf(a[j], c[j], -(b[j] * s[j]));    // explicit contraction: the fused forms (gemm256 epilogue, attention backward)
      ob[j] = fmaf(b[j], c[j], a[j] * s[j]);       // use the same two expressions and stay bit-identical to this kernel
    }
    *(u32x4_t*)p = pack8(oa);
    *(u32x4_t*)(p + half) = pack8(ob);
  }
}

// ---------------------------------------------------------------------------------------------------------
// SwiGLU (reference: modeling_llama.py:211-242): act = silu(gate) * up, gate = gu[:, :I], up = gu[:, I:]
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act,
                                                         long long rows, int I) {
  const int cpr = I >> 3;
  const long long total = rows * cpr;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long r = idx / cpr;
    const int c = (int)(idx % cpr);
    float g[8], u[8], o[8];
    unpack8(*(const u32x4_t*)(gu + r * 2 * I + c * 8), g);
    unpack8(*(const u32x4_t*)(gu + r * 2 * I + I + c * 8), u);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = swiglu_fwd_elem(g[j], u[j]);
    *(u32x4_t*)(act + r * I + c * 8) = pack8(o);
  }
}

// dgu[:, :I] = dact * up * silu'(gate); dgu[:, I:] = dact * silu(gate); optionally re-materialises act.
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ dact, const bf16_t* __restrict__ gu,
                                                         bf16_t* __restrict__ dgu, bf16_t* __restrict__ act_out,
                                                         long long rows, int I) {
  const int cpr = I >> 3;
  const long long total = rows * cpr;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long r = idx / cpr;
    const int c = (int)(idx % cpr);
    float g[8], u[8], d[8], dg[8], du[8], a[8];
    unpack8(*(const u32x4_t*)(gu + r * 2 * I + c * 8), g);
    unpack8(*(const u32x4_t*)(gu + r * 2 * I + I + c * 8), u);
    unpack8(*(const u32x4_t*)(dact + r * I + c * 8), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = mla_sigmoid(g[j]);
      const float sl = g[j] * s;
      dg[j] = d[j] * u[j] * (s + sl * (1.f - s));
      du[j] = d[j] * sl;
      a[j] = sl * u[j];
    }
    *(u32x4_t*)(dgu + r * 2 * I + c * 8) = pack8(dg);
    *(u32x4_t*)(dgu + r * 2 * I + I + c * 8) = pack8(du);
    if (act_out) *(u32x4_t*)(act_out + r * I + c * 8) = pack8(a);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Activations of the small heads: 0 GELU(erf) (MLPProjector util/nn_utils.py:21-34, MLP_GELU), 1 GELU(tanh)
// (timm Mlp in ActionEmbedder/FinalLayer, models/diffusion/models.py:112-123,173-189), 2 ReLU (contrastive heads,
// models/mla/fuser/contrastive.py:173-183), 3 SiLU (TimestepEmbedder, models/diffusion/models.py:34-38).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_f(int kind, float x) {
  switch (kind) {
    case 0: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    case 1: { const float t = tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)); return 0.5f * x * (1.f + t); }
    case 2: return x > 0.f ? x : 0.f;
    default: return x / (1.f + __expf(-x));
  }
}
__device__ __forceinline__ float act_df(int kind, float x) {
  switch (kind) {
    case 0: return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
    case 1: {
      const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
      const float t = tanhf(u);
      return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * 0.7978845608028654f * (1.f + 3.f * 0.044715f * x * x);
    }
    case 2: return x > 0.f ? 1.f : 0.f;
    default: { const float s = mla_sigmoid(x); return s + x * s * (1.f - s); }
  }
}
__global__ __launch_bounds__(256) void act_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long long n,
                                                      int kind) {
  const long long nch = n >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nch; i += (long long)gridDim.x * 256) {
    float v[8];
    unpack8(*(const u32x4_t*)(x + i * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = act_f(kind, v[j]);
    *(u32x4_t*)(y + i * 8) = pack8(v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long long i = (nch << 3) + threadIdx.x;
    y[i] = f2bf(act_f(kind, bf2f(x[i])));
  }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                      bf16_t* __restrict__ dx, long long n, int kind) {
  const long long nch = n >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nch; i += (long long)gridDim.x * 256) {
    float v[8], d[8];
    unpack8(*(const u32x4_t*)(x + i * 8), v);
    unpack8(*(const u32x4_t*)(dy + i * 8), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] *= act_df(kind, v[j]);
    *(u32x4_t*)(dx + i * 8) = pack8(d);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long long i = (nch << 3) + threadIdx.x;
    dx[i] = f2bf(bf2f(dy[i]) * act_df(kind, bf2f(x[i])));
  }
}

// ---------------------------------------------------------------------------------------------------------
// casts, adds, embedding
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long long n) {
  const long long nch = n >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nch; i += (long long)gridDim.x * 256) {
    const f32x4_t a = *(const f32x4_t*)(x + i * 8), b = *(const f32x4_t*)(x + i * 8 + 4);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    *(u32x4_t*)(y + i * 8) = pack8(v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) y[(nch << 3) + threadIdx.x] = f2bf(x[(nch << 3) + threadIdx.x]);
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long long n) {
  const long long nch = n >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nch; i += (long long)gridDim.x * 256) {
    float v[8];
    unpack8(*(const u32x4_t*)(x + i * 8), v);
    *(f32x4_t*)(y + i * 8) = f32x4_t{v[0], v[1], v[2], v[3]};
    *(f32x4_t*)(y + i * 8 + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) y[(nch << 3) + threadIdx.x] = bf2f(x[(nch << 3) + threadIdx.x]);
}
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                       bf16_t* __restrict__ y, long long n) {
  const long long nch = n >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nch; i += (long long)gridDim.x * 256) {
    float u[8], v[8];
    unpack8(*(const u32x4_t*)(a + i * 8), u);
    unpack8(*(const u32x4_t*)(b + i * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) u[j] += v[j];
    *(u32x4_t*)(y + i * 8) = pack8(u);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long long i = (nch << 3) + threadIdx.x;
    y[i] = f2bf(bf2f(a[i]) + bf2f(b[i]));
  }
}

// out[t][:] = table[ids[t]][:]   (LlamaModel.embed_tokens, modeling_llama.py:975-976)
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const long long* __restrict__ ids, const bf16_t* __restrict__ table,
                                                            bf16_t* __restrict__ out, long long tokens, int H, int vocab) {
  const int cpr = H >> 3;
  const long long total = tokens * cpr;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const long long t = idx / cpr;
    const int c = (int)(idx % cpr);
    long long id = ids[t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    *(u32x4_t*)(out + t * H + c * 8) = *(const u32x4_t*)(table + id * H + c * 8);
  }
}
// grad[ids[t]][h] += dy[t][h]  (autograd of nn.Embedding, LlamaModel.embed_tokens, modeling_llama.py:975-976; the resized table
// has no padding_idx, models/backbones/llm/llama2.py:77, so the <PAD> row accumulates like any other).
// Deterministic without atomics and without a sort: every element is accumulated in ascending token order by exactly one thread.
// Workgroup (x, y) owns the vocabulary slice y; its waves own 256 columns each (4 per lane) and share one scan of the ids: each
// wave loads 64 ids of a stage, keeps the ones inside the slice and posts them in LDS; every wave then walks the matching tokens
// of the whole stage (ballot of the posted ids) with the dy loads of up to EMB_AHEAD matches in flight before the first is
// consumed. A run of equal ids -- the padding at the end of a sequence -- stays in registers and costs loads of dy only.
// One workgroup cannot pull more than ~50 GB/s, so a long run is pre-reduced by many: embedding_run_reduce_kernel sums every aligned
// batch of 64 tokens that carries ONE id into an fp32 workspace row (ascending token order), and the walk then consumes that batch
// as a single row. Both kernels evaluate the same predicate on the same ids, so no flags are exchanged.
// Round 1 walked every token in 16 workgroups: 0.5 ms at 16 Ki tokens, 23 ms at 64 Ki (tools/bench_embedding.py).
#define EMB_AHEAD 16
#define EMB_MAX_WAVES 16
__global__ __launch_bounds__(256) void embedding_run_reduce_kernel(const long long* __restrict__ ids, const bf16_t* __restrict__ dy,
                                                                   float* __restrict__ ws, int H, int vocab) {
  const long long c = blockIdx.x;  // batch of tokens [64 c, 64 c + 64), all inside the input
  const long long idl = ids[c * 64 + (threadIdx.x & 63)];
  const long long id0 = __shfl(idl, 0, 64);
  if (id0 < 0 || id0 >= vocab || __ballot(idl == id0) != ~0ull) return;
  const bf16_t* src = dy + c * 64 * H;
  float* dst = ws + c * H;
  for (int h = threadIdx.x * 4; h < H; h += 1024) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int r = 0; r < 64; r += 16) {
      u32x2_t w[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) w[u] = *(const u32x2_t*)(src + (long long)(r + u) * H + h);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        acc[0] += bflo(w[u][0]); acc[1] += bfhi(w[u][0]);
        acc[2] += bflo(w[u][1]); acc[3] += bfhi(w[u][1]);
      }
    }
    *(f32x4_t*)(dst + h) = acc;
  }
}
template <bool RAGGED>  // RAGGED: H is not a multiple of 256 (per-lane column predicate)
__global__ __launch_bounds__(64 * EMB_MAX_WAVES) void embedding_bwd_kernel(const long long* __restrict__ ids,
                                                                           const bf16_t* __restrict__ dy, float* __restrict__ grad,
                                                                           const float* __restrict__ ws, long long tokens, int H,
                                                                           int vocab) {
  __shared__ int ids_s[2][64 * EMB_MAX_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int hw = (blockIdx.x * nw + wave) * 256, h = hw + lane * 4;  // H % 4 == 0 (checked by the launcher)
  const bool live_w = hw < H, live = RAGGED ? h < H : true;
  const int v0 = (int)((long long)vocab * blockIdx.y / gridDim.y), v1 = (int)((long long)vocab * (blockIdx.y + 1) / gridDim.y);
  const int stage = nw * 64;
  int cur = -1, buf = 0;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  long long idn = wave * 64 + lane < tokens ? ids[wave * 64 + lane] : -1;
  for (long long t0 = 0; t0 < tokens; t0 += stage, buf ^= 1) {
    ids_s[buf][wave * 64 + lane] = (idn >= v0 && idn < v1) ? (int)idn : -1;
    const long long tn = t0 + stage + wave * 64 + lane;
    idn = tn < tokens ? ids[tn] : -1;
    __syncthreads();  // one barrier per stage: the other buffer is rewritten only after everyone passed this one
    if (!live_w) continue;
    for (int b = 0; b < nw; ++b) {
      const int id = ids_s[buf][b * 64 + lane];
      unsigned long long mask = __ballot(id >= 0);
      if (ws && mask == ~0ull) {
        const int id0 = __builtin_amdgcn_readfirstlane(id);
        if (__ballot(id == id0) == ~0ull) {  // the whole batch is one id: its sum is in the workspace
          if (live) {
            const f32x4_t r = *(const f32x4_t*)(ws + ((t0 >> 6) + b) * H + h);
            if (id0 != cur) {
              if (cur >= 0) *(f32x4_t*)(grad + (long long)cur * H + h) = acc;
              acc = *(const f32x4_t*)(grad + (long long)id0 * H + h);
            }
            acc += r;
          }
          cur = id0;
          continue;
        }
      }
      const bf16_t* dyb = dy + (t0 + b * 64) * H + h;
      while (mask) {
        u32x2_t w[EMB_AHEAD];
        unsigned long long m2 = mask;
#pragma unroll
        for (int u = 0; u < EMB_AHEAD; ++u) {
          if (m2) {
            const int k = __ffsll((long long)m2) - 1;
            m2 &= m2 - 1;
            if (live) w[u] = *(const u32x2_t*)(dyb + (long long)k * H);
          }
        }
#pragma unroll
        for (int u = 0; u < EMB_AHEAD; ++u) {
          if (mask) {
            const int k = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int idk = __builtin_amdgcn_readlane(id, k);
            if (idk != cur && live) {
              if (cur >= 0) *(f32x4_t*)(grad + (long long)cur * H + h) = acc;
              acc = *(const f32x4_t*)(grad + (long long)idk * H + h);
            }
            cur = idk;
            acc[0] += bflo(w[u][0]); acc[1] += bfhi(w[u][0]);
            acc[2] += bflo(w[u][1]); acc[3] += bfhi(w[u][1]);
          }
        }
      }
    }
  }
  if (cur >= 0 && live_w && live) *(f32x4_t*)(grad + (long long)cur * H + h) = acc;
}

// ---------------------------------------------------------------------------------------------------------
// Optimizer: fused AdamW over the local fp32 shard, also refreshes the bf16 compute copy
// (reference: torch.optim.AdamW built at training/strategies/fsdp.py:257; clip at :308-310)
// ---------------------------------------------------------------------------------------------------------
// One element of the update with every contraction spelled out: written as plain a * b + c the compiler is free to fuse either
// product, and it chooses differently in differently shaped copies of the loop (measured: the two-group launch differed from the
// one-group launch in the last bit of 24 % of the moments). decay = 1 - lr * wd or 1.
__device__ __forceinline__ void adamw_elem(float& p, float g, float& m, float& v, float gs, float decay, float beta1, float beta2,
                                           float step, float isq, float eps) {
  const float gg = g * gs;
  const float mm = fmaf(beta1, m, (1.f - beta1) * gg);
  const float vv = fmaf(beta2, v, ((1.f - beta2) * gg) * gg);
  const float den = fmaf(sqrtf(vv), isq, eps);
  p = fmaf(-step, mm / den, p * decay);
  m = mm;
  v = vv;
}
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ p16, long long n, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                                                    const float* __restrict__ grad_scale, long long n_decay) {
  const float gs = grad_scale ? *grad_scale : 1.f;
  const float step = lr / bc1, isq = 1.f / sqrtf(bc2);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float pp = p[i], mm = m[i], vv = v[i];
    adamw_elem(pp, g[i], mm, vv, gs, i < n_decay ? 1.f - lr * wd : 1.f, beta1, beta2, step, isq, eps);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (p16) p16[i] = f2bf(pp);
  }
}

// same update, 4 elements per lane per trip (16-B accesses on the seven fp32 streams, 8-B on the bf16 copy): the update is
// element-wise, so the results are bit-identical to adamw_kernel; n4 = number of whole float4 groups.
// UNR = float4 groups per lane and trip (all their loads issued before the first use); NT = non-temporal accesses.
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void adamw_vec4_kernel(f32x4_t* __restrict__ p, const f32x4_t* __restrict__ g, f32x4_t* __restrict__ m,
                                                         f32x4_t* __restrict__ v, u32x2_t* __restrict__ p16, long long n4, float lr,
                                                         float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                                                         const float* __restrict__ grad_scale, long long n4_decay) {
  const float gs = grad_scale ? *grad_scale : 1.f;
  const float step = lr / bc1, isq = 1.f / sqrtf(bc2), decay_on = 1.f - lr * wd;
  const long long chunk = ((((n4) + gridDim.x - 1) / gridDim.x) + 256 * UNR - 1) / (256 * UNR) * (256 * UNR);
  const long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
  for (long long i0 = lo + threadIdx.x; i0 < hi; i0 += 256 * UNR) {
    f32x4_t g4[UNR], p4[UNR], m4[UNR], v4[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long long i = i0 + u * 256;
      if (i < hi) {
        if (NT) {
          g4[u] = __builtin_nontemporal_load(g + i); p4[u] = __builtin_nontemporal_load(p + i);
          m4[u] = __builtin_nontemporal_load(m + i); v4[u] = __builtin_nontemporal_load(v + i);
        } else {
          g4[u] = g[i]; p4[u] = p[i]; m4[u] = m[i]; v4[u] = v[i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long long i = i0 + u * 256;
      if (i >= hi) continue;
      const float decay = i < n4_decay ? decay_on : 1.f;     // groups [0, n4_decay) are weight-decayed, the rest (norms, biases) not
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pp = p4[u][r], mm = m4[u][r], vv = v4[u][r];
        adamw_elem(pp, g4[u][r], mm, vv, gs, decay, beta1, beta2, step, isq, eps);
        p4[u][r] = pp; m4[u][r] = mm; v4[u][r] = vv;
      }
      // every stream is touched exactly once per step: non-temporal loads and stores (+3 % stand-alone, tools/bench_adamw.py)
      if (NT) {
        __builtin_nontemporal_store(p4[u], p + i); __builtin_nontemporal_store(m4[u], m + i); __builtin_nontemporal_store(v4[u], v + i);
      } else {
        p[i] = p4[u]; m[i] = m4[u]; v[i] = v4[u];
      }
      if (p16) {
        u32x2_t o;
        o[0] = pack2bf(p4[u][0], p4[u][1]);
        o[1] = pack2bf(p4[u][2], p4[u][3]);
        if (NT) __builtin_nontemporal_store(o, p16 + i); else p16[i] = o;
      }
    }
  }
}

// partial[blockIdx] = sum of squares of this block's slice (deterministic two-stage grad-norm); 16-B loads
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, long long n) {
  __shared__ float scratch[16];
  float s0 = 0.f, s1 = 0.f;
  const long long n4 = n >> 2;
  const f32x4_t* x4 = (const f32x4_t*)x;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4_t v = x4[i];
    s0 += v[0] * v[0] + v[1] * v[1];
    s1 += v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[(n4 << 2) + threadIdx.x]; s0 += v * v; }
  const float s = block_sum(s0 + s1, scratch);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// out[0] (+)= sum(partial[0..P))
__global__ __launch_bounds__(256) void sum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int P,
                                                        int accumulate) {
  __shared__ float scratch[16];
  float s = 0.f;
  int i0 = 0;
  if ((((uintptr_t)partial) & 15) == 0) {   // 16-B loads, four in flight per lane: a 200 K-entry arena of wgrad partials in ~30 us, not ~400
    const f32x4_t* p4 = (const f32x4_t*)partial;
    const int n4 = P >> 2;
    f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int i = threadIdx.x;
    for (; i + 768 < n4; i += 1024) { a0 += p4[i]; a1 += p4[i + 256]; a2 += p4[i + 512]; a3 += p4[i + 768]; }
    for (; i < n4; i += 256) a0 += p4[i];
    const f32x4_t a = (a0 + a1) + (a2 + a3);
    s = (a[0] + a[1]) + (a[2] + a[3]);
    i0 = n4 << 2;
  }
  for (int i = i0 + threadIdx.x; i < P; i += 256) s += partial[i];
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}
// coef[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)); norm_out[0] = sqrt(sumsq[0])   (torch clip_grad_norm_)
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ coef,
                                 float* __restrict__ norm_out) {
  const float nrm = sqrtf(sumsq[0]);
  const float c = max_norm / (nrm + 1e-6f);
  coef[0] = c < 1.f ? c : 1.f;
  if (norm_out) norm_out[0] = nrm;
}

// x_t = sqrt_ac[t]*x0 + sqrt_1mac[t]*noise   (GaussianDiffusion.q_sample, models/diffusion/gaussian_diffusion.py:214-229)
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const long long* __restrict__ t, const float* __restrict__ sqrt_ac,
                                                       const float* __restrict__ sqrt_1mac, float* __restrict__ out,
                                                       int batch, int per, int nsteps) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= batch * per) return;
  long long tt = t[i / per];
  tt = tt < 0 ? 0 : (tt >= nsteps ? nsteps - 1 : tt);
  out[i] = sqrt_ac[tt] * x0[i] + sqrt_1mac[tt] * noise[i];
}

// out[r][:] = src[idx[r]][:]  /  dst[idx[r]][:] = src[r][:]   (sequence assembly; idx is injective for the scatter)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ src, const long long* __restrict__ idx,
                                                          bf16_t* __restrict__ out, long long rows, int H, int scatter) {
  const int cpr = H >> 3;
  const long long total = rows * cpr;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / cpr;
    const int c = (int)(e % cpr);
    const long long s = idx[r];
    if (scatter) *(u32x4_t*)(out + s * H + c * 8) = *(const u32x4_t*)(src + r * H + c * 8);
    else *(u32x4_t*)(out + r * H + c * 8) = *(const u32x4_t*)(src + s * H + c * 8);
  }
}

inline int grid_for(long long work_items, int cap = 4096) {
  long long b = (work_items + 255) / 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace

#define AL16(p) ((((uintptr_t)(p)) & 15) == 0)

extern "C" int mla_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int H, float eps,
                               hipStream_t stream) {
  MLA_CHECK_ARG(x && w && y, "mla_rmsnorm_fwd: null pointer");
  MLA_CHECK_ARG(rows > 0 && H > 0 && H % 8 == 0 && H <= 8192, "mla_rmsnorm_fwd: need H%%8==0, H<=8192 (H=%d)", H);
  MLA_CHECK_ARG(AL16(x) && AL16(w) && AL16(y), "mla_rmsnorm_fwd: pointers must be 16-B aligned");
  hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3(rows), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y,
                     rstd, rows, H, eps);
  MLA_LAUNCH_CHECK();
}

// xg = bf16(x * w) and (optional) rstd [rows] of LlamaRMSNorm: the two halves of a folded RMSNorm (see rmsnorm_prep_kernel)
extern "C" int mla_rmsnorm_prep(const void* x, const void* w, void* xg, float* rstd, int rows, int H, float eps, hipStream_t stream) {
  MLA_CHECK_ARG(x && w && xg, "mla_rmsnorm_prep: null pointer");
  MLA_CHECK_ARG(rows > 0 && H > 0 && H % 8 == 0 && H <= 8192, "mla_rmsnorm_prep: need H%%8==0, H<=8192 (H=%d)", H);
  MLA_CHECK_ARG(AL16(x) && AL16(w) && AL16(xg), "mla_rmsnorm_prep: pointers must be 16-B aligned");
  hipLaunchKernelGGL(rmsnorm_prep_kernel, dim3(rows), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)xg, rstd, rows,
                     H, eps);
  MLA_LAUNCH_CHECK();
}

// workspace: nblocks*H floats (nblocks = mla_rmsnorm_bwd_blocks(rows)); dw may be null (frozen weight)
extern "C" int mla_rmsnorm_bwd_blocks(int rows) {
  // 512 = two 4-wave workgroups per CU. More (1024: round-2 first half) only adds concurrent row streams and dw partials:
  // 107 vs 130 us stand-alone at 17 536 x 4096, -2.1 ms/step in three same-box pairs (256: 152 us, 768: 116, 2048: 128)
  static const int cap = getenv("MLA_RMSNORM_BWD_BLOCKS") ? atoi(getenv("MLA_RMSNORM_BWD_BLOCKS")) : 512;
  return rows < cap ? rows : cap;
}
extern "C" int mla_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                               float* dw, int dw_accumulate, int rows, int H, float* workspace, size_t workspace_bytes,
                               hipStream_t stream) {
  MLA_CHECK_ARG(dy && x && w && rstd && dx, "mla_rmsnorm_bwd: null pointer");
  MLA_CHECK_ARG(rows > 0 && H > 0 && H % 8 == 0 && H <= 8192, "mla_rmsnorm_bwd: need H%%8==0, H<=8192 (H=%d)", H);
  const int nb = mla_rmsnorm_bwd_blocks(rows);
  MLA_CHECK_ARG(!dw || (workspace && workspace_bytes >= (size_t)nb * H * sizeof(float)), "mla_rmsnorm_bwd: workspace too small");
  // register arrays sized for the row length: H <= 4096 needs 2 chunks of 8 per thread (half the VGPRs -> twice the resident waves)
  if (H <= 4096)
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<2>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x,
                       (const bf16_t*)w, rstd, (const bf16_t*)dres, (bf16_t*)dx, dw ? workspace : nullptr, rows, H);
  else
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<NORM_MAXC>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x,
                       (const bf16_t*)w, rstd, (const bf16_t*)dres, (bf16_t*)dx, dw ? workspace : nullptr, rows, H);
  if (dw)
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((H + RP_COLS - 1) / RP_COLS), dim3(256), 0, stream, workspace, dw, nb, H, dw_accumulate);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_timm_rmsnorm_fwd(const void* x, const void* w, void* y, float* mean, float* rstd, int rows, int H, float eps,
                                    hipStream_t stream) {
  MLA_CHECK_ARG(x && w && y && mean && rstd, "mla_timm_rmsnorm_fwd: null pointer");
  MLA_CHECK_ARG(rows > 0 && H > 8 && H % 8 == 0 && H <= 8192, "mla_timm_rmsnorm_fwd: need H%%8==0, 8<H<=8192 (H=%d)", H);
  MLA_CHECK_ARG(AL16(x) && AL16(w) && AL16(y), "mla_timm_rmsnorm_fwd: pointers must be 16-B aligned");
  hipLaunchKernelGGL(timm_rmsnorm_fwd_kernel, dim3(rows), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y,
                     mean, rstd, rows, H, eps);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_timm_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, void* dx,
                                    float* dw, int dw_accumulate, int rows, int H, float* workspace, size_t workspace_bytes,
                                    hipStream_t stream) {
  MLA_CHECK_ARG(dy && x && w && mean && rstd && dx, "mla_timm_rmsnorm_bwd: null pointer");
  MLA_CHECK_ARG(rows > 0 && H > 8 && H % 8 == 0 && H <= 8192, "mla_timm_rmsnorm_bwd: need H%%8==0, 8<H<=8192 (H=%d)", H);
  const int nb = mla_rmsnorm_bwd_blocks(rows);
  MLA_CHECK_ARG(!dw || (workspace && workspace_bytes >= (size_t)nb * H * sizeof(float)), "mla_timm_rmsnorm_bwd: workspace too small");
  hipLaunchKernelGGL(timm_rmsnorm_bwd_kernel, dim3(nb), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x,
                     (const bf16_t*)w, mean, rstd, (bf16_t*)dx, dw ? workspace : nullptr, rows, H);
  if (dw)
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((H + RP_COLS - 1) / RP_COLS), dim3(256), 0, stream, workspace, dw, nb, H, dw_accumulate);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_colsum_blocks(int rows) { int r = rows / 256; return r < 1 ? 1 : (r > 64 ? 64 : r); }
// out[n] (+)= sum_r dy[r][n];  workspace: mla_colsum_blocks(rows)*N floats
extern "C" int mla_colsum_bf16(const void* dy, float* out, int accumulate, int rows, int N, int ld, float* workspace,
                               size_t workspace_bytes, hipStream_t stream) {
  MLA_CHECK_ARG(dy && out && workspace, "mla_colsum_bf16: null pointer");
  const int rs = mla_colsum_blocks(rows);
  MLA_CHECK_ARG(workspace_bytes >= (size_t)rs * N * sizeof(float), "mla_colsum_bf16: workspace too small");
  if ((N & 7) == 0 && (ld & 7) == 0 && AL16(dy))
    hipLaunchKernelGGL(colsum_partial_vec_kernel, dim3((N + 63) / 64, rs), dim3(256), 0, stream, (const bf16_t*)dy, workspace, rows, N, ld);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 63) / 64, rs), dim3(256), 0, stream, (const bf16_t*)dy, workspace, rows, N, ld);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((N + RP_COLS - 1) / RP_COLS), dim3(256), 0, stream, workspace, out, rs, N, accumulate);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_layernorm_fwd(const void* x, const void* w, const void* b, void* y, int rows, int H, float eps,
                                 hipStream_t stream) {
  MLA_CHECK_ARG(x && w && b && y, "mla_layernorm_fwd: null pointer");
  MLA_CHECK_ARG(rows > 0 && H % 8 == 0 && H <= 8192, "mla_layernorm_fwd: need H%%8==0, H<=8192");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(rows), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)w,
                     (const bf16_t*)b, (bf16_t*)y, rows, H, eps);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_rope_inplace(void* buf, const float* cos_t, const float* sin_t, long long tokens, int S, int nheads,
                                int D, int ld, int q_off, int k_off, int backward, hipStream_t stream) {
  MLA_CHECK_ARG(buf && cos_t && sin_t, "mla_rope_inplace: null pointer");
  MLA_CHECK_ARG(D % 16 == 0 && ld % 8 == 0 && q_off % 8 == 0 && k_off % 8 == 0 && S > 0, "mla_rope_inplace: bad layout");
  MLA_CHECK_ARG(AL16(buf) && AL16(cos_t) && AL16(sin_t), "mla_rope_inplace: alignment");
  const long long total = tokens * 2 * nheads * (D / 16);
  hipLaunchKernelGGL(rope_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, stream, (bf16_t*)buf, cos_t, sin_t, tokens, S,
                     nheads, D, ld, q_off, k_off, backward ? -1.f : 1.f);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_swiglu_fwd(const void* gu, void* act, long long rows, int I, hipStream_t stream) {
  MLA_CHECK_ARG(gu && act && I % 8 == 0, "mla_swiglu_fwd: bad args");
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(rows * (I / 8), 8192)), dim3(256), 0, stream, (const bf16_t*)gu,
                     (bf16_t*)act, rows, I);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_swiglu_bwd(const void* dact, const void* gu, void* dgu, void* act_out, long long rows, int I,
                              hipStream_t stream) {
  MLA_CHECK_ARG(dact && gu && dgu && I % 8 == 0, "mla_swiglu_bwd: bad args");
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(rows * (I / 8), 8192)), dim3(256), 0, stream, (const bf16_t*)dact,
                     (const bf16_t*)gu, (bf16_t*)dgu, (bf16_t*)act_out, rows, I);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_act_fwd(const void* x, void* y, long long n, int kind, hipStream_t stream) {
  MLA_CHECK_ARG(x && y && kind >= 0 && kind <= 3 && AL16(x) && AL16(y), "mla_act_fwd: bad args");
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, n, kind);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_act_bwd(const void* dy, const void* x, void* dx, long long n, int kind, hipStream_t stream) {
  MLA_CHECK_ARG(dy && x && dx && kind >= 0 && kind <= 3 && AL16(x) && AL16(dy) && AL16(dx), "mla_act_bwd: bad args");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x,
                     (bf16_t*)dx, n, kind);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_cast_f32_to_bf16(const float* x, void* y, long long n, hipStream_t stream) {
  MLA_CHECK_ARG(x && y && AL16(x) && AL16(y), "mla_cast_f32_to_bf16: bad args");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, stream, x, (bf16_t*)y, n);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_cast_bf16_to_f32(const void* x, float* y, long long n, hipStream_t stream) {
  MLA_CHECK_ARG(x && y && AL16(x) && AL16(y), "mla_cast_bf16_to_f32: bad args");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, stream, (const bf16_t*)x, y, n);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_add_bf16(const void* a, const void* b, void* y, long long n, hipStream_t stream) {
  MLA_CHECK_ARG(a && b && y && AL16(a) && AL16(b) && AL16(y), "mla_add_bf16: bad args");
  hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(n / 8 + 1)), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b,
                     (bf16_t*)y, n);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_embedding_fwd(const long long* ids, const void* table, void* out, long long tokens, int H, int vocab,
                                 hipStream_t stream) {
  MLA_CHECK_ARG(ids && table && out && H % 8 == 0, "mla_embedding_fwd: bad args");
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for(tokens * (H / 8))), dim3(256), 0, stream, ids, (const bf16_t*)table,
                     (bf16_t*)out, tokens, H, vocab);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_embedding_bwd(const long long* ids, const void* dy, float* grad, float* workspace, long long tokens, int H,
                                 int vocab, hipStream_t stream) {
  MLA_CHECK_ARG(ids && dy && grad && H % 4 == 0 && vocab > 0, "mla_embedding_bwd: bad args (H %% 4 == 0)");
  if (tokens <= 0) return 0;
  // vocabulary slices: every workgroup scans all ids (the same addresses at the same time: the scan is bound by one L2 channel and
  // costs ~ slices x tokens), every slice walks ~tokens / slices dependent load -> add -> store steps. 256 measured best at
  // 16 Ki and 64 Ki tokens.
  static const int env_vp = getenv("MLA_EMB_SLICES") ? atoi(getenv("MLA_EMB_SLICES")) : 0;
  long long vp = env_vp > 0 ? env_vp : (tokens + 63) / 64;
  vp = vp < 1 ? 1 : (vp > 256 && env_vp <= 0 ? 256 : vp);
  if (vp > vocab) vp = vocab;
  if (workspace && tokens >= 64)
    hipLaunchKernelGGL(embedding_run_reduce_kernel, dim3((unsigned)(tokens / 64)), dim3(256), 0, stream, ids, (const bf16_t*)dy,
                       workspace, H, vocab);
  const int ncb = (H + 255) / 256, nw = ncb < EMB_MAX_WAVES ? ncb : EMB_MAX_WAVES;
  const dim3 grid((ncb + nw - 1) / nw, (unsigned)vp), block(64 * nw);
  if (H % 256 == 0)
    hipLaunchKernelGGL(embedding_bwd_kernel<false>, grid, block, 0, stream, ids, (const bf16_t*)dy, grad, workspace, tokens, H, vocab);
  else
    hipLaunchKernelGGL(embedding_bwd_kernel<true>, grid, block, 0, stream, ids, (const bf16_t*)dy, grad, workspace, tokens, H, vocab);
  MLA_LAUNCH_CHECK();
}

static int adamw_grid_cap() {
  static const int cap = getenv("MLA_ADAMW_BLOCKS") ? atoi(getenv("MLA_ADAMW_BLOCKS")) : 8192;
  return cap;
}
static int adamw_impl(float* p, const float* g, float* m, float* v, void* p16, long long n, long long n_decay, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int step, const float* grad_scale, hipStream_t stream) {
  MLA_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1 && n_decay >= 0 && n_decay <= n, "mla_adamw_step: bad args");
  if (n == 0) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  // (the decay boundary must not cut a 16-B group: otherwise everything goes through the scalar kernel)
  const long long n4 = (AL16(p) && AL16(g) && AL16(m) && AL16(v) && (p16 == nullptr || ((uintptr_t)p16 & 7) == 0) && (n_decay & 3) == 0) ? n / 4 : 0;
  if (n4) {
    static const int variant = getenv("MLA_ADAMW_VARIANT") ? atoi(getenv("MLA_ADAMW_VARIANT")) : 0;   // A/B only (tools/bench_adamw.py)
    const dim3 grid(grid_for(n4, adamw_grid_cap()));
#define ADAMW_LAUNCH(U, N) hipLaunchKernelGGL((adamw_vec4_kernel<U, N>), grid, dim3(256), 0, stream, (f32x4_t*)p, (const f32x4_t*)g, (f32x4_t*)m, \
                       (f32x4_t*)v, (u32x2_t*)p16, n4, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, n_decay / 4)
    // default: four float4 groups per lane and trip, non-temporal (stand-alone 6.20-6.29 TB/s vs 5.99-6.15 with one group, same box,
    // alternating; plain accesses 5.9-6.0) -- tools/ab_adamw.sh
    if (variant == 1) ADAMW_LAUNCH(2, true);
    else if (variant == 9) ADAMW_LAUNCH(1, true);     // rounds 2-3
    else if (variant == 3) ADAMW_LAUNCH(1, false);
    else if (variant == 4) ADAMW_LAUNCH(2, false);
    else ADAMW_LAUNCH(4, true);
#undef ADAMW_LAUNCH
  }
  const long long done = n4 * 4;
  if (done < n)
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n - done, 8192)), dim3(256), 0, stream, p + done, g + done, m + done, v + done,
                       p16 ? (bf16_t*)p16 + done : (bf16_t*)nullptr, n - done, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale,
                       n_decay > done ? n_decay - done : 0);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_adamw_step(float* p, const float* g, float* m, float* v, void* p16, long long n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, const float* grad_scale,
                              hipStream_t stream) {
  return adamw_impl(p, g, m, v, p16, n, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, stream);
}
// One launch for a flat parameter range laid out [weight-decayed | not decayed] (FlatUnit: matrices first, then norm weights and
// biases): elements [0, n_decay) get `weight_decay`, the rest none -- the reference's two AdamW parameter groups
// (training/strategies/fsdp.py:231-257) without a second launch per unit.
extern "C" int mla_adamw_step_groups(float* p, const float* g, float* m, float* v, void* p16, long long n, long long n_decay, float lr,
                                     float beta1, float beta2, float eps, float weight_decay, int step, const float* grad_scale,
                                     hipStream_t stream) {
  return adamw_impl(p, g, m, v, p16, n, n_decay, lr, beta1, beta2, eps, weight_decay, step, grad_scale, stream);
}

// out[0] (+)= sum(x^2); workspace >= 2048 floats
extern "C" int mla_sumsq_f32(const float* x, long long n, float* out, int accumulate, float* workspace, size_t workspace_bytes,
                             hipStream_t stream) {
  MLA_CHECK_ARG(x && out && workspace && workspace_bytes >= 2048 * sizeof(float) && AL16(x), "mla_sumsq_f32: bad args");
  const int nb = grid_for(n / 4 + 1, 2048);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, stream, x, workspace, n);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, stream, workspace, out, nb, accumulate);
  MLA_LAUNCH_CHECK();
}
// out[0] (+)= sum(partial[0 .. n)) in a fixed order (one workgroup): the second stage of the gradient norm when the first stage was
// produced elsewhere (sum-of-squares partials of the wgrad GEMM epilogues, mla_gemm_bf16_ws_sq)
extern "C" int mla_sum_partials(const float* partial, int n, float* out, int accumulate, hipStream_t stream) {
  MLA_CHECK_ARG(partial && out && n >= 0, "mla_sum_partials: bad args");
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, stream, partial, out, n, accumulate);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out, hipStream_t stream) {
  MLA_CHECK_ARG(sumsq && coef, "mla_clip_coef: bad args");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, stream, sumsq, max_norm, coef, norm_out);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_q_sample(const float* x0, const float* noise, const long long* t, const float* sqrt_ac,
                            const float* sqrt_1mac, float* out, int batch, int per, int nsteps, hipStream_t stream) {
  MLA_CHECK_ARG(x0 && noise && t && sqrt_ac && sqrt_1mac && out, "mla_q_sample: null pointer");
  hipLaunchKernelGGL(q_sample_kernel, dim3((batch * per + 255) / 256), dim3(256), 0, stream, x0, noise, t, sqrt_ac, sqrt_1mac,
                     out, batch, per, nsteps);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_gather_rows_bf16(const void* src, const long long* idx, void* out, long long rows, int H, int scatter,
                                    hipStream_t stream) {
  MLA_CHECK_ARG(src && idx && out && H % 8 == 0 && AL16(src) && AL16(out), "mla_gather_rows_bf16: bad args");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(rows * (H / 8))), dim3(256), 0, stream, (const bf16_t*)src, idx, (bf16_t*)out,
                     rows, H, scatter);
  MLA_LAUNCH_CHECK();
}
