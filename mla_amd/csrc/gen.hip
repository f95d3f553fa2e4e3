// Kernels of the post-training generation heads (BASELINE config[3]; reference: models/mla/generation/models.py,
// gen_loss.py, utils.py and PrismaticVLM.compute_generation_losses models/vlm/prismatic.py:771-838).
// The heads are torch nn.TransformerDecoder stacks (8 heads x 512) + a PointNet-style decoder; their matmuls run on the
// MFMA GEMMs (batched per (sample, head) for the attention products); this file holds the HBM-bound rest.
#include "common.h"
#include <math.h>

namespace {

// counter-based RNG for dropout: same (seed, index) -> same decision in forward and backward
__device__ __forceinline__ uint32_t hash_u32(unsigned long long seed, unsigned long long idx) {
  unsigned long long x = idx * 0x9E3779B97F4A7C15ull + seed;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull;
  x ^= x >> 32;
  return (uint32_t)x;
}
__device__ __forceinline__ float keep_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
  if (p <= 0.f) return 1.f;
  return ((hash_u32(seed, idx) >> 8) * (1.0f / 16777216.0f)) >= p ? inv_keep : 0.f;
}

// ---- softmax over the first nvalid of ncols columns (+ dropout); one block per row
// P (pre-dropout, bf16) is kept for backward, Pd = dropout(P) feeds the P.V product
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const float* __restrict__ s, bf16_t* __restrict__ P, bf16_t* __restrict__ Pd,
                                                               int ncols, int nvalid, float p, unsigned long long seed) {
  __shared__ float scratch[16];
  const long long r = blockIdx.x;
  const float* row = s + r * ncols;
  float m = -INFINITY;
  for (int c = threadIdx.x; c < nvalid; c += 256) m = fmaxf(m, row[c]);
  m = block_max(m, scratch);
  float sum = 0.f;
  for (int c = threadIdx.x; c < nvalid; c += 256) sum += __expf(row[c] - m);
  sum = block_sum(sum, scratch);
  const float inv = 1.f / sum, ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int c = threadIdx.x; c < ncols; c += 256) {
    const float pr = c < nvalid ? __expf(row[c] - m) * inv : 0.f;
    P[r * ncols + c] = f2bf(pr);
    Pd[r * ncols + c] = f2bf(pr * keep_scale(seed, (unsigned long long)(r * ncols + c), p, ik));
  }
}
// dS = P' * (g - sum_c(g * P')),  g = dPd * mask / (1 - p),  P' = P / sum_c(P)
// Round 6: the saved probabilities are bf16, so they do not sum to one -- and for a near-uniform row (random-init cross-attention over
// ~550 keys) every P_c = 1 / n rounds THE SAME WAY, leaving sum(P) off by up to 2^-9. Then sum_c(dS_c) = (1 - sum P) * dot != 0, an
// error that is perfectly aligned with the MEAN key and therefore survives dQ = dS K where the signal (aligned with key differences)
// cancels: measured 12 % on the query-projection gradient of the tactile decoder's first cross-attention against 2.3 % for the
// reference in bf16 autocast, whose softmax backward runs on fp32 probabilities (tests/parity_util.py, profiles/r6_parity_table.txt).
// Renormalising the bf16 values in fp32 restores sum_c(dS_c) = 0 (before dS is rounded) at the cost of one more term in the reduction.
// The incoming dPd = dO V^T is fp32 in the product path (round 6): torch's fused attention -- what nn.MultiheadAttention runs under the
// reference's bf16 autocast -- never rounds dP, and for one query over ~550 near-uniformly weighted keys dP_k = dO . v_k is a large
// common value plus a small per-key variation, of which only the VARIATION survives (dP_k - dot): rounded to bf16 it left 12 % error on
// the query-projection gradient of the tactile decoder's first cross-attention (reference bf16: 2.3 %; same table).
__device__ __forceinline__ float dpd_load(const float* p, long long i) { return p[i]; }
__device__ __forceinline__ float dpd_load(const bf16_t* p, long long i) { return bf2f(p[i]); }
template <typename TG>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const TG* __restrict__ dPd, const bf16_t* __restrict__ P,
                                                               bf16_t* __restrict__ dS, int ncols, int nvalid, float p,
                                                               unsigned long long seed) {
  __shared__ float scratch[16];
  const long long r = blockIdx.x;
  const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float dot = 0.f, psum = 0.f;
  for (int c = threadIdx.x; c < nvalid; c += 256) {
    const float g = dpd_load(dPd, r * ncols + c) * keep_scale(seed, (unsigned long long)(r * ncols + c), p, ik);
    const float pr = bf2f(P[r * ncols + c]);
    dot += g * pr;
    psum += pr;
  }
  dot = block_sum(dot, scratch);
  psum = block_sum(psum, scratch);
  const float inv = psum > 0.f ? 1.f / psum : 0.f;
  dot *= inv;
  for (int c = threadIdx.x; c < ncols; c += 256) {
    float v = 0.f;
    if (c < nvalid) {
      const float g = dpd_load(dPd, r * ncols + c) * keep_scale(seed, (unsigned long long)(r * ncols + c), p, ik);
      v = bf2f(P[r * ncols + c]) * inv * (g - dot);
    }
    dS[r * ncols + c] = f2bf(v);
  }
}

// ---- y = residual + dropout(x) (residual optional) ; dx = dy * mask / (1-p)
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res, bf16_t* __restrict__ y,
                                                          long long n, float p, unsigned long long seed) {
  const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = bf2f(x[i]) * keep_scale(seed, (unsigned long long)i, p, ik);
    y[i] = f2bf(res ? v + bf2f(res[i]) : v);
  }
}
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, long long n, float p,
                                                          unsigned long long seed) {
  const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    dx[i] = f2bf(bf2f(dy[i]) * keep_scale(seed, (unsigned long long)i, p, ik));
}
// y[b][...] = x[b][...] * scale[b]   (DropPath: per-sample keep / (1 - p))
__global__ __launch_bounds__(256) void scale_batch_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale, bf16_t* __restrict__ y,
                                                          long long per, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    y[i] = f2bf(bf2f(x[i]) * scale[i / per]);
}

// ---- LayerNorm with saved statistics + backward (two-stage weight/bias gradients). H % 8 == 0, H <= 8192
constexpr int LN_MAXC = 4;
__global__ __launch_bounds__(256) void layernorm_stats_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                  const bf16_t* __restrict__ b, bf16_t* __restrict__ y,
                                                                  float* __restrict__ mean_o, float* __restrict__ rstd_o, int H, float eps) {
  __shared__ float scratch[16];
  const long long row = blockIdx.x;
  const int nchunk = H >> 3;
  float xv[LN_MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      const u32x4_t v = *(const u32x4_t*)(x + row * H + ch * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) { xv[c][2 * j] = bflo(v[j]); xv[c][2 * j + 1] = bfhi(v[j]); s += xv[c][2 * j] + xv[c][2 * j + 1]; }
    }
  }
  const float mean = block_sum(s, scratch) / (float)H;
  float vs = 0.f;
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = xv[c][j] - mean; vs += d * d; }
  }
  const float rstd = 1.0f / sqrtf(block_sum(vs, scratch) / (float)H + eps);
  if (threadIdx.x == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      const u32x4_t wv = *(const u32x4_t*)(w + ch * 8), bv = *(const u32x4_t*)(b + ch * 8);
      u32x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = pack2bf((xv[c][2 * j] - mean) * rstd * bflo(wv[j]) + bflo(bv[j]), (xv[c][2 * j + 1] - mean) * rstd * bfhi(wv[j]) + bfhi(bv[j]));
      *(u32x4_t*)(y + row * H + ch * 8) = o;
    }
  }
}
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w ; partial[blk][0][h] = sum dy*xhat, partial[blk][1][h] = sum dy
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                            const bf16_t* __restrict__ w, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                            float* __restrict__ partial, long long rows, int H) {
  __shared__ float scratch[16];
  const int nchunk = H >> 3;
  float wv[LN_MAXC][8], dwa[LN_MAXC][8], dba[LN_MAXC][8];
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
#pragma unroll
    for (int j = 0; j < 8; ++j) { dwa[c][j] = 0.f; dba[c][j] = 0.f; wv[c][j] = 0.f; }
    if (ch < nchunk) {
      const u32x4_t v = *(const u32x4_t*)(w + ch * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) { wv[c][2 * j] = bflo(v[j]); wv[c][2 * j + 1] = bfhi(v[j]); }
    }
  }
  for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
    const float mu = mean[row], rs = rstd[row];
    float xh[LN_MAXC][8], g[LN_MAXC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = threadIdx.x + c * 256;
      if (ch < nchunk) {
        const u32x4_t xv = *(const u32x4_t*)(x + row * H + ch * 8), dv = *(const u32x4_t*)(dy + row * H + ch * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xx = (j & 1) ? bfhi(xv[j >> 1]) : bflo(xv[j >> 1]);
          const float dd = (j & 1) ? bfhi(dv[j >> 1]) : bflo(dv[j >> 1]);
          xh[c][j] = (xx - mu) * rs;
          g[c][j] = dd * wv[c][j];
          s1 += g[c][j];
          s2 += g[c][j] * xh[c][j];
          dwa[c][j] += dd * xh[c][j];
          dba[c][j] += dd;
        }
      }
    }
    s1 = block_sum(s1, scratch) / (float)H;
    s2 = block_sum(s2, scratch) / (float)H;
#pragma unroll
    for (int c = 0; c < LN_MAXC; ++c) {
      const int ch = threadIdx.x + c * 256;
      if (ch < nchunk) {
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          o[j] = pack2bf(rs * (g[c][2 * j] - s1 - xh[c][2 * j] * s2), rs * (g[c][2 * j + 1] - s1 - xh[c][2 * j + 1] * s2));
        *(u32x4_t*)(dx + row * H + ch * 8) = o;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < LN_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      float* o = partial + ((size_t)blockIdx.x * 2) * H + ch * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) { o[j] = dwa[c][j]; o[H + j] = dba[c][j]; }
    }
  }
}
// out[n] (+)= sum_p partial[p*stride + n]
__global__ __launch_bounds__(256) void reduce_strided_kernel(const float* __restrict__ partial, float* __restrict__ out, int P, int N,
                                                             long long stride, int accumulate) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  float s = 0.f;
  if (n < N)
    for (int pp = rl; pp < P; pp += 4) s += partial[(size_t)pp * stride + n];
  red[rl][c] = s;
  __syncthreads();
  if (rl == 0 && n < N) {
    const float t = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    out[n] = accumulate ? out[n] + t : t;
  }
}

// ---- mean over the sequence: y[b][c] = mean_s x[b][s][c] ; backward broadcasts dy / S
__global__ __launch_bounds__(256) void seqmean_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int S, int C) {
  const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int t = 0; t < S; ++t) s += bf2f(x[((size_t)b * S + t) * C + c]);
  y[(size_t)b * C + c] = f2bf(s / (float)S);
}
__global__ __launch_bounds__(256) void seqmean_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int S, int C, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long b = i / ((long long)S * C);
    const int c = (int)(i % C);
    dx[i] = f2bf(bf2f(dy[b * C + c]) / (float)S);
  }
}

// ---- BatchNorm (train) backward. partial[blk][0][c] = sum dy, [1][c] = sum dy * xhat over the block's rows
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ var,
                                                             float* __restrict__ partial, long long rows, int C, float eps) {
  __shared__ float s1[4][64], s2[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int ch = blockIdx.x * 64 + c;
  const long long per = (rows + gridDim.y - 1) / gridDim.y;
  const long long r0 = blockIdx.y * per, r1 = (r0 + per) < rows ? (r0 + per) : rows;
  float a = 0.f, q = 0.f;
  if (ch < C) {
    const float mu = mean[ch], rs = 1.0f / sqrtf(var[ch] + eps);
    for (long long r = r0 + rl; r < r1; r += 4) {
      const float d = bf2f(dy[r * C + ch]);
      a += d;
      q += d * (bf2f(x[r * C + ch]) - mu) * rs;
    }
  }
  s1[rl][c] = a; s2[rl][c] = q;
  __syncthreads();
  if (rl == 0 && ch < C) {
    partial[((size_t)blockIdx.y * 2) * C + ch] = s1[0][c] + s1[1][c] + s1[2][c] + s1[3][c];
    partial[((size_t)blockIdx.y * 2 + 1) * C + ch] = s2[0][c] + s2[1][c] + s2[2][c] + s2[3][c];
  }
}
// dx = w * rstd * (dy - sum_dy / N - xhat * sum_dyxhat / N)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ var,
                                                           const bf16_t* __restrict__ w, const float* __restrict__ sum_dy,
                                                           const float* __restrict__ sum_dyxh, bf16_t* __restrict__ dx, long long rows, int C,
                                                           float eps) {
  const long long n = rows * C;
  const float invn = 1.f / (float)rows;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float rs = 1.0f / sqrtf(var[c] + eps);
    const float xh = (bf2f(x[i]) - mean[c]) * rs;
    dx[i] = f2bf(bf2f(w[c]) * rs * (bf2f(dy[i]) - sum_dy[c] * invn - xh * sum_dyxh[c] * invn));
  }
}

// ---- chamfer_distance_l2 (gen_loss.py:12-18): Euclidean cdist, min over each axis, means. fp32 points.
// d1[b][n] = min_m |pred_n - gt_m| (idx1), d2[b][m] = min_n |pred_n - gt_m| (idx2)
__global__ __launch_bounds__(256) void chamfer_min_kernel(const float* __restrict__ a, const float* __restrict__ bpts, float* __restrict__ dmin,
                                                          int* __restrict__ imin, int Na, int Nb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sb = (float*)smem;
  const int b = blockIdx.y;
  for (int e = threadIdx.x; e < Nb * 3; e += 256) sb[e] = bpts[(size_t)b * Nb * 3 + e];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Na) return;
  const float x = a[((size_t)b * Na + i) * 3], y = a[((size_t)b * Na + i) * 3 + 1], z = a[((size_t)b * Na + i) * 3 + 2];
  float best = INFINITY;
  int bi = 0;
  for (int j = 0; j < Nb; ++j) {
    const float dx = x - sb[j * 3], dy = y - sb[j * 3 + 1], dz = z - sb[j * 3 + 2];
    const float d = dx * dx + dy * dy + dz * dz;
    if (d < best) { best = d; bi = j; }
  }
  dmin[(size_t)b * Na + i] = sqrtf(best);
  imin[(size_t)b * Na + i] = bi;
}
// dpred[b][n] = g/B * ( (p_n - gt[idx1[n]]) / d1[n] / N  +  sum_{m: idx2[m]==n} (p_n - gt_m) / d2[m] / M )
__global__ __launch_bounds__(256) void chamfer_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                          const float* __restrict__ d1, const int* __restrict__ i1,
                                                          const float* __restrict__ d2, const int* __restrict__ i2,
                                                          const float* __restrict__ gscale, float* __restrict__ dpred, int B, int N, int M) {
  const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float g = gscale[0] / (float)B;
  const float* p = pred + ((size_t)b * N + n) * 3;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  {
    const float* q = gt + ((size_t)b * M + i1[(size_t)b * N + n]) * 3;
    const float d = fmaxf(d1[(size_t)b * N + n], 1e-12f), s = g / ((float)N * d);
    gx += (p[0] - q[0]) * s; gy += (p[1] - q[1]) * s; gz += (p[2] - q[2]) * s;
  }
  for (int m = 0; m < M; ++m) {
    if (i2[(size_t)b * M + m] == n) {
      const float* q = gt + ((size_t)b * M + m) * 3;
      const float d = fmaxf(d2[(size_t)b * M + m], 1e-12f), s = g / ((float)M * d);
      gx += (p[0] - q[0]) * s; gy += (p[1] - q[1]) * s; gz += (p[2] - q[2]) * s;
    }
  }
  float* o = dpred + ((size_t)b * N + n) * 3;
  o[0] = gx; o[1] = gy; o[2] = gz;
}

// ---- image generation loss with the full ROI (use_roi = False): generated = 0.05 * curr + 5 * tanh(delta_raw)
// (ImageGenerationModule._generate_generated_patches models.py:226-286 with an all-true mask), patches addressed straight
// in the [B, 3|4, 672, 672] images (images_to_patches utils.py:7-18). sums[0] += (pred-gt)^2, [1] += |pred-gt|, [2] += |delta|
template <typename TI>
__device__ __forceinline__ float img_at(const TI* img, int CT, int HW, int ps, int b, int patch, int e) {
  const int g = HW / ps;
  const int c = e / (ps * ps), rem = e % (ps * ps), yy = rem / ps, xx = rem % ps;
  const int py = patch / g, px = patch % g;
  const size_t off = (((size_t)b * CT + c) * HW + (py * ps + yy)) * HW + (px * ps + xx);
  return sizeof(TI) == 4 ? ((const float*)img)[off] : bf2f(((const bf16_t*)img)[off]);
}
template <typename TI>
__global__ __launch_bounds__(256) void imgloss_fwd_kernel(const bf16_t* __restrict__ draw, const TI* __restrict__ curr, const TI* __restrict__ next,
                                                          float* __restrict__ partial, int B, int CTc, int CTn, int HW, int ps, int npatch,
                                                          float clip, int ld) {
  __shared__ float scratch[16];
  const int pd = 3 * ps * ps;
  const long long n = (long long)B * npatch * pd;
  float a = 0.f, l1 = 0.f, dl = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int e = (int)(i % pd);
    const long long r = i / pd;
    const int patch = (int)(r % npatch), b = (int)(r / npatch);
    const float delta = clip * tanhf(bf2f(draw[r * ld + e]));
    const float pred = 0.05f * img_at<TI>(curr, CTc, HW, ps, b, patch, e) + delta;
    const float diff = pred - img_at<TI>(next, CTn, HW, ps, b, patch, e);
    a += diff * diff; l1 += fabsf(diff); dl += fabsf(delta);
  }
  a = block_sum(a, scratch); l1 = block_sum(l1, scratch); dl = block_sum(dl, scratch);
  if (threadIdx.x == 0) { partial[blockIdx.x * 3] = a; partial[blockIdx.x * 3 + 1] = l1; partial[blockIdx.x * 3 + 2] = dl; }
}
// d(draw) = gscale * [ 2 diff / n + 0.5 sign(diff) / n - 0.1 sign(delta) / n ] * clip * (1 - tanh^2)
template <typename TI>
__global__ __launch_bounds__(256) void imgloss_bwd_kernel(const bf16_t* __restrict__ draw, const TI* __restrict__ curr, const TI* __restrict__ next,
                                                          const float* __restrict__ gscale, bf16_t* __restrict__ ddraw, int B, int CTc, int CTn,
                                                          int HW, int ps, int npatch, float clip, int ld) {
  const int pd = 3 * ps * ps;
  const long long n = (long long)B * npatch * pd;
  const float g = gscale[0] / (float)n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int e = (int)(i % pd);
    const long long r = i / pd;
    const int patch = (int)(r % npatch), b = (int)(r / npatch);
    const float th = tanhf(bf2f(draw[r * ld + e]));
    const float delta = clip * th;
    const float diff = 0.05f * img_at<TI>(curr, CTc, HW, ps, b, patch, e) + delta - img_at<TI>(next, CTn, HW, ps, b, patch, e);
    const float sg = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f), sd = delta > 0.f ? 1.f : (delta < 0.f ? -1.f : 0.f);
    ddraw[r * ld + e] = f2bf(g * (2.f * diff + 0.5f * sg - 0.1f * sd) * clip * (1.f - th * th));
  }
}
// ---- image generation with a partial ROI (use_roi = True): ImageGenerationModule._generate_generated_patches models.py:226-286
// + the three image loss terms of compute_generation_losses prismatic.py:780-816. One block per patch (b, p):
//   ROI patch:     pred = 0.05 * curr + delta                                   -> MSE + 0.5 L1 over ROI elements
//   other patches: pred = alpha * (warp(curr; tx, ty) + delta) + (1 - alpha) * curr -> 0.01 * L1 over background elements
//   all patches:   -0.1 * mean |delta|
// delta = clip * tanh(draw), alpha = sigmoid(a_raw), (tx, ty) = shift * tanh(o_raw); warp = bilinear sample of the 42x42 patch at
// (x + tx, y + ty) with border clamp (affine_grid + grid_sample(align_corners = True, padding_mode = 'border') of a pure translation).
// partial[blk][0..4] = { sum diff^2 (ROI), sum |diff| (ROI), sum |diff| (bg), sum |delta|, 0 }
template <typename TI>
__device__ __forceinline__ float patch_px(const TI* img, int CT, int HW, int b, int c, int y, int x) {
  const size_t off = (((size_t)b * CT + c) * HW + y) * HW + x;
  return sizeof(TI) == 4 ? ((const float*)img)[off] : bf2f(((const bf16_t*)img)[off]);
}
// bilinear sample inside the patch whose top-left pixel is (y0, x0); returns value and d/dx, d/dy (zero where the border clamp acts)
template <typename TI>
__device__ __forceinline__ float warp_sample(const TI* img, int CT, int HW, int ps, int b, int c, int y0, int x0, float sy, float sx,
                                             float& ddx, float& ddy) {
  const float mx = (float)(ps - 1);
  const bool cx = sx < 0.f || sx > mx, cy = sy < 0.f || sy > mx;
  sx = fminf(fmaxf(sx, 0.f), mx);
  sy = fminf(fmaxf(sy, 0.f), mx);
  int xi = (int)floorf(sx), yi = (int)floorf(sy);
  const float wx = sx - (float)xi, wy = sy - (float)yi;
  const int xj = xi + 1 < ps ? xi + 1 : ps - 1, yj = yi + 1 < ps ? yi + 1 : ps - 1;
  const float p00 = patch_px<TI>(img, CT, HW, b, c, y0 + yi, x0 + xi), p01 = patch_px<TI>(img, CT, HW, b, c, y0 + yi, x0 + xj);
  const float p10 = patch_px<TI>(img, CT, HW, b, c, y0 + yj, x0 + xi), p11 = patch_px<TI>(img, CT, HW, b, c, y0 + yj, x0 + xj);
  ddx = cx ? 0.f : ((1.f - wy) * (p01 - p00) + wy * (p11 - p10));
  ddy = cy ? 0.f : ((1.f - wx) * (p10 - p00) + wx * (p11 - p01));
  return (1.f - wy) * ((1.f - wx) * p00 + wx * p01) + wy * ((1.f - wx) * p10 + wx * p11);
}
// MODE 0: forward partial sums; MODE 1: backward (ddraw per element, d a_raw / d o_raw per patch)
template <typename TI, int MODE>
__global__ __launch_bounds__(256) void imgroi_kernel(const bf16_t* __restrict__ draw, int ld, const bf16_t* __restrict__ araw, int lda_,
                                                     const bf16_t* __restrict__ oraw, int ldo, const unsigned char* __restrict__ roi,
                                                     const TI* __restrict__ curr, const TI* __restrict__ next, float* __restrict__ partial,
                                                     const float* __restrict__ coef, bf16_t* __restrict__ ddraw,
                                                     float* __restrict__ dalpha_raw, float* __restrict__ doff_raw, int CTc, int CTn, int HW,
                                                     int ps, int npatch, float clip, float shift) {
  __shared__ float scratch[16];
  const int bp = blockIdx.x, b = bp / npatch, patch = bp % npatch;
  const int g = HW / ps, py = patch / g, px = patch % g, y0 = py * ps, x0 = px * ps;
  const bool in_roi = roi[bp] != 0;
  const float alpha = 1.f / (1.f + __expf(-bf2f(araw[(size_t)bp * lda_])));
  const float ttx = tanhf(bf2f(oraw[(size_t)bp * ldo])), tty = tanhf(bf2f(oraw[(size_t)bp * ldo + 1]));
  const float tx = shift * ttx, ty = shift * tty;
  const int pd = 3 * ps * ps;
  // coef (MODE 1) = { g / n_roi, g / n_bg * 0.01, g * (-0.1) / n_all } -- zero where a term is absent
  const float k_roi = MODE ? coef[0] : 0.f, k_bg = MODE ? coef[1] : 0.f, k_dl = MODE ? coef[2] : 0.f;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, ga = 0.f, gx = 0.f, gy = 0.f;
  for (int e = threadIdx.x; e < pd; e += 256) {
    const int c = e / (ps * ps), rem = e % (ps * ps), yy = rem / ps, xx = rem % ps;
    const float th = tanhf(bf2f(draw[(size_t)bp * ld + e]));
    const float delta = clip * th;
    const float cv = patch_px<TI>(curr, CTc, HW, b, c, y0 + yy, x0 + xx);
    const float nv = patch_px<TI>(next, CTn, HW, b, c, y0 + yy, x0 + xx);
    float gdelta;
    if (in_roi) {
      const float diff = 0.05f * cv + delta - nv;
      s0 += diff * diff; s1 += fabsf(diff);
      gdelta = k_roi * (2.f * diff + 0.5f * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)));
    } else {
      float ddx, ddy;
      const float wv = warp_sample<TI>(curr, CTc, HW, ps, b, c, y0, x0, (float)yy + ty, (float)xx + tx, ddx, ddy);
      const float diff = alpha * (wv + delta) + (1.f - alpha) * cv - nv;
      s2 += fabsf(diff);
      const float gp = k_bg * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
      gdelta = gp * alpha;
      ga += gp * (wv + delta - cv);
      gx += gp * alpha * ddx;
      gy += gp * alpha * ddy;
    }
    s3 += fabsf(delta);
    if (MODE) {
      gdelta += k_dl * (delta > 0.f ? 1.f : (delta < 0.f ? -1.f : 0.f));
      ddraw[(size_t)bp * ld + e] = f2bf(gdelta * clip * (1.f - th * th));
    }
  }
  if (MODE == 0) {
    s0 = block_sum(s0, scratch); s1 = block_sum(s1, scratch); s2 = block_sum(s2, scratch); s3 = block_sum(s3, scratch);
    if (threadIdx.x == 0) { float* o = partial + (size_t)bp * 4; o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; }
  } else {
    ga = block_sum(ga, scratch); gx = block_sum(gx, scratch); gy = block_sum(gy, scratch);
    if (threadIdx.x == 0) {
      dalpha_raw[bp] = ga * alpha * (1.f - alpha);
      doff_raw[(size_t)bp * 2] = gx * shift * (1.f - ttx * ttx);
      doff_raw[(size_t)bp * 2 + 1] = gy * shift * (1.f - tty * tty);
    }
    for (int e = pd + threadIdx.x; e < ld; e += 256) ddraw[(size_t)bp * ld + e] = 0;      // padding columns
  }
}
__global__ __launch_bounds__(256) void sum4_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int P) {
  __shared__ float scratch[16];
  for (int k = 0; k < 4; ++k) {
    float s = 0.f;
    for (int i = threadIdx.x; i < P; i += 256) s += partial[(size_t)i * 4 + k];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) out[k] = s;
  }
}

__global__ __launch_bounds__(256) void sum3_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int P) {
  __shared__ float scratch[16];
  for (int k = 0; k < 3; ++k) {
    float s = 0.f;
    for (int i = threadIdx.x; i < P; i += 256) s += partial[i * 3 + k];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) out[k] = s;
  }
}
__global__ __launch_bounds__(256) void mean_f32_kernel(const float* __restrict__ x, float* __restrict__ out, long long n) {
  __shared__ float scratch[16];
  float s = 0.f;
  for (long long i = threadIdx.x; i < n; i += 256) s += x[i];
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}

inline int gridn(long long items, int cap = 8192) {
  long long b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int mla_softmax_rows_fwd(const float* scores, void* P, void* Pd, long long rows, int ncols, int nvalid, float p,
                                    unsigned long long seed, hipStream_t stream) {
  MLA_CHECK_ARG(scores && P && Pd && rows > 0 && nvalid > 0 && nvalid <= ncols && p >= 0.f && p < 1.f, "mla_softmax_rows_fwd: bad args");
  hipLaunchKernelGGL(softmax_rows_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, stream, scores, (bf16_t*)P, (bf16_t*)Pd, ncols, nvalid, p, seed);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_softmax_rows_bwd(const void* dPd, int dpd_fp32, const void* P, void* dS, long long rows, int ncols, int nvalid, float p,
                                    unsigned long long seed, hipStream_t stream) {
  MLA_CHECK_ARG(dPd && P && dS && rows > 0 && nvalid <= ncols, "mla_softmax_rows_bwd: bad args");
  if (dpd_fp32)
    hipLaunchKernelGGL(softmax_rows_bwd_kernel<float>, dim3((unsigned)rows), dim3(256), 0, stream, (const float*)dPd, (const bf16_t*)P, (bf16_t*)dS,
                       ncols, nvalid, p, seed);
  else
    hipLaunchKernelGGL(softmax_rows_bwd_kernel<bf16_t>, dim3((unsigned)rows), dim3(256), 0, stream, (const bf16_t*)dPd, (const bf16_t*)P, (bf16_t*)dS,
                       ncols, nvalid, p, seed);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_dropout_fwd(const void* x, const void* residual, void* y, long long n, float p, unsigned long long seed, hipStream_t stream) {
  MLA_CHECK_ARG(x && y && n > 0 && p >= 0.f && p < 1.f, "mla_dropout_fwd: bad args");
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(gridn(n)), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, n, p, seed);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_dropout_bwd(const void* dy, void* dx, long long n, float p, unsigned long long seed, hipStream_t stream) {
  MLA_CHECK_ARG(dy && dx && n > 0, "mla_dropout_bwd: bad args");
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(gridn(n)), dim3(256), 0, stream, (const bf16_t*)dy, (bf16_t*)dx, n, p, seed);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_scale_batch(const void* x, const float* scale, void* y, long long batch, long long per, hipStream_t stream) {
  MLA_CHECK_ARG(x && scale && y && batch > 0 && per > 0, "mla_scale_batch: bad args");
  hipLaunchKernelGGL(scale_batch_kernel, dim3(gridn(batch * per)), dim3(256), 0, stream, (const bf16_t*)x, scale, (bf16_t*)y, per, batch * per);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_layernorm_stats_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, long long rows, int H,
                                       float eps, hipStream_t stream) {
  MLA_CHECK_ARG(x && w && b && y && mean && rstd && rows > 0 && H % 8 == 0 && H <= 8192, "mla_layernorm_stats_fwd: need H%%8==0, H<=8192");
  hipLaunchKernelGGL(layernorm_stats_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b,
                     (bf16_t*)y, mean, rstd, H, eps);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_layernorm_bwd_blocks(long long rows) { return (int)(rows < 512 ? rows : 512); }
// workspace >= blocks * 2 * H floats; dw / db may be null (then only dx is produced)
extern "C" int mla_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, void* dx, float* dw,
                                 float* db, int accumulate, long long rows, int H, float* workspace, size_t workspace_bytes,
                                 hipStream_t stream) {
  MLA_CHECK_ARG(dy && x && w && mean && rstd && dx && workspace && rows > 0 && H % 8 == 0 && H <= 8192, "mla_layernorm_bwd: bad args");
  const int nb = mla_layernorm_bwd_blocks(rows);
  MLA_CHECK_ARG(workspace_bytes >= (size_t)nb * 2 * H * sizeof(float), "mla_layernorm_bwd: workspace too small");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nb), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)w, mean, rstd,
                     (bf16_t*)dx, workspace, rows, H);
  if (dw) hipLaunchKernelGGL(reduce_strided_kernel, dim3((H + 63) / 64), dim3(256), 0, stream, workspace, dw, nb, H, 2LL * H, accumulate);
  if (db) hipLaunchKernelGGL(reduce_strided_kernel, dim3((H + 63) / 64), dim3(256), 0, stream, workspace + H, db, nb, H, 2LL * H, accumulate);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_seqmean_fwd(const void* x, void* y, int B, int S, int C, hipStream_t stream) {
  MLA_CHECK_ARG(x && y && B > 0 && S > 0 && C > 0, "mla_seqmean_fwd: bad args");
  hipLaunchKernelGGL(seqmean_fwd_kernel, dim3((C + 255) / 256, B), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, S, C);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_seqmean_bwd(const void* dy, void* dx, int B, int S, int C, hipStream_t stream) {
  MLA_CHECK_ARG(dy && dx, "mla_seqmean_bwd: bad args");
  const long long n = (long long)B * S * C;
  hipLaunchKernelGGL(seqmean_bwd_kernel, dim3(gridn(n)), dim3(256), 0, stream, (const bf16_t*)dy, (bf16_t*)dx, S, C, n);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_bn_bwd_blocks(long long rows) { long long r = rows / 1024; return (int)(r < 1 ? 1 : (r > 128 ? 128 : r)); }
// workspace >= blocks*2*C + 2*C floats. dw/db (+)= sums; dx = BatchNorm(train) input gradient
extern "C" int mla_bn_bwd(const void* dy, const void* x, const float* mean, const float* var, const void* w, void* dx, float* dw, float* db,
                          int accumulate, long long rows, int C, float eps, float* workspace, size_t workspace_bytes, hipStream_t stream) {
  MLA_CHECK_ARG(dy && x && mean && var && w && dx && workspace && rows > 0, "mla_bn_bwd: bad args");
  const int P = mla_bn_bwd_blocks(rows);
  MLA_CHECK_ARG(workspace_bytes >= ((size_t)P * 2 * C + 2 * C) * sizeof(float), "mla_bn_bwd: workspace too small");
  float* sums = workspace + (size_t)P * 2 * C;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3((C + 63) / 64, P), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, mean, var, workspace,
                     rows, C, eps);
  hipLaunchKernelGGL(reduce_strided_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, workspace, sums, P, C, 2LL * C, 0);
  hipLaunchKernelGGL(reduce_strided_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, workspace + C, sums + C, P, C, 2LL * C, 0);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gridn(rows * C)), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)x, mean, var,
                     (const bf16_t*)w, sums, sums + C, (bf16_t*)dx, rows, C, eps);
  if (db) hipLaunchKernelGGL(reduce_strided_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, sums, db, 1, C, (long long)C, accumulate);
  if (dw) hipLaunchKernelGGL(reduce_strided_kernel, dim3((C + 63) / 64), dim3(256), 0, stream, sums + C, dw, 1, C, (long long)C, accumulate);
  MLA_LAUNCH_CHECK();
}
// loss[0] = mean_b( mean_n d1 + mean_m d2 ); d1/i1 [B,N], d2/i2 [B,M] kept for backward. N, M <= 4096
extern "C" int mla_chamfer_fwd(const float* pred, const float* gt, float* d1, int* i1, float* d2, int* i2, float* loss, int B, int N, int M,
                               float* workspace, size_t workspace_bytes, hipStream_t stream) {
  MLA_CHECK_ARG(pred && gt && d1 && i1 && d2 && i2 && loss && workspace && N <= 4096 && M <= 4096, "mla_chamfer_fwd: bad args (N, M <= 4096)");
  MLA_CHECK_ARG(workspace_bytes >= 2 * sizeof(float), "mla_chamfer_fwd: workspace too small");
  hipLaunchKernelGGL(chamfer_min_kernel, dim3((N + 255) / 256, B), dim3(256), M * 3 * sizeof(float), stream, pred, gt, d1, i1, N, M);
  hipLaunchKernelGGL(chamfer_min_kernel, dim3((M + 255) / 256, B), dim3(256), N * 3 * sizeof(float), stream, gt, pred, d2, i2, M, N);
  hipLaunchKernelGGL(mean_f32_kernel, dim3(1), dim3(256), 0, stream, d1, workspace, (long long)B * N);
  hipLaunchKernelGGL(mean_f32_kernel, dim3(1), dim3(256), 0, stream, d2, workspace + 1, (long long)B * M);
  hipLaunchKernelGGL(reduce_strided_kernel, dim3(1), dim3(256), 0, stream, workspace, loss, 2, 1, 1LL, 0);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_chamfer_bwd(const float* pred, const float* gt, const float* d1, const int* i1, const float* d2, const int* i2,
                               const float* gscale, float* dpred, int B, int N, int M, hipStream_t stream) {
  MLA_CHECK_ARG(pred && gt && d1 && i1 && d2 && i2 && gscale && dpred, "mla_chamfer_bwd: null pointer");
  hipLaunchKernelGGL(chamfer_bwd_kernel, dim3((N + 255) / 256, B), dim3(256), 0, stream, pred, gt, d1, i1, d2, i2, gscale, dpred, B, N, M);
  MLA_LAUNCH_CHECK();
}
// sums[3] = { sum (pred-gt)^2, sum |pred-gt|, sum |delta| }; workspace >= 2048*3 floats. delta_raw rows have pitch ld >= 3*ps*ps
// (the delta head's output is padded to the MFMA tile granularity); backward writes only the valid columns
extern "C" int mla_imgloss_fwd(const void* delta_raw, int ld, const void* curr, const void* next, int img_fp32, float* sums, int B,
                               int CT_curr, int CT_next, int HW, int ps, float clip, float* workspace, size_t workspace_bytes,
                               hipStream_t stream) {
  MLA_CHECK_ARG(delta_raw && curr && next && sums && workspace && HW % ps == 0 && ld >= 3 * ps * ps, "mla_imgloss_fwd: bad args");
  const int nb = 2048;
  MLA_CHECK_ARG(workspace_bytes >= (size_t)nb * 3 * sizeof(float), "mla_imgloss_fwd: workspace too small");
  const int npatch = (HW / ps) * (HW / ps);
  if (img_fp32)
    hipLaunchKernelGGL(imgloss_fwd_kernel<float>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)delta_raw, (const float*)curr, (const float*)next,
                       workspace, B, CT_curr, CT_next, HW, ps, npatch, clip, ld);
  else
    hipLaunchKernelGGL(imgloss_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, stream, (const bf16_t*)delta_raw, (const bf16_t*)curr,
                       (const bf16_t*)next, workspace, B, CT_curr, CT_next, HW, ps, npatch, clip, ld);
  hipLaunchKernelGGL(sum3_final_kernel, dim3(1), dim3(256), 0, stream, workspace, sums, nb);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_imgloss_bwd(const void* delta_raw, int ld, const void* curr, const void* next, int img_fp32, const float* gscale,
                               void* ddelta_raw, int B, int CT_curr, int CT_next, int HW, int ps, float clip, hipStream_t stream) {
  MLA_CHECK_ARG(delta_raw && curr && next && gscale && ddelta_raw, "mla_imgloss_bwd: null pointer");
  const int npatch = (HW / ps) * (HW / ps);
  const long long n = (long long)B * npatch * 3 * ps * ps;
  if (img_fp32)
    hipLaunchKernelGGL(imgloss_bwd_kernel<float>, dim3(gridn(n)), dim3(256), 0, stream, (const bf16_t*)delta_raw, (const float*)curr,
                       (const float*)next, gscale, (bf16_t*)ddelta_raw, B, CT_curr, CT_next, HW, ps, npatch, clip, ld);
  else
    hipLaunchKernelGGL(imgloss_bwd_kernel<bf16_t>, dim3(gridn(n)), dim3(256), 0, stream, (const bf16_t*)delta_raw, (const bf16_t*)curr,
                       (const bf16_t*)next, gscale, (bf16_t*)ddelta_raw, B, CT_curr, CT_next, HW, ps, npatch, clip, ld);
  MLA_LAUNCH_CHECK();
}

// sums[4] = { sum diff^2 over ROI elements, sum |diff| over ROI, sum |diff| over background, sum |delta| over all };
// roi [B * npatch] bytes (non-zero = patch inside the dilated ROI); a_raw / o_raw are the alpha / offset head outputs (row pitches
// lda / ldo elements, columns 0 and 0..1 used); workspace >= B * npatch * 4 floats
extern "C" int mla_imgroi_fwd(const void* delta_raw, int ld, const void* a_raw, int lda, const void* o_raw, int ldo,
                              const unsigned char* roi, const void* curr, const void* next, int img_fp32, float* sums, int B, int CT_curr,
                              int CT_next, int HW, int ps, float clip, float shift, float* workspace, size_t workspace_bytes,
                              hipStream_t stream) {
  MLA_CHECK_ARG(delta_raw && a_raw && o_raw && roi && curr && next && sums && workspace && HW % ps == 0 && ld >= 3 * ps * ps && ldo >= 2,
                "mla_imgroi_fwd: bad args");
  const int npatch = (HW / ps) * (HW / ps);
  MLA_CHECK_ARG(workspace_bytes >= (size_t)B * npatch * 4 * sizeof(float), "mla_imgroi_fwd: workspace too small");
  if (img_fp32)
    hipLaunchKernelGGL((imgroi_kernel<float, 0>), dim3(B * npatch), dim3(256), 0, stream, (const bf16_t*)delta_raw, ld, (const bf16_t*)a_raw, lda,
                       (const bf16_t*)o_raw, ldo, roi, (const float*)curr, (const float*)next, workspace, nullptr, nullptr, nullptr, nullptr,
                       CT_curr, CT_next, HW, ps, npatch, clip, shift);
  else
    hipLaunchKernelGGL((imgroi_kernel<bf16_t, 0>), dim3(B * npatch), dim3(256), 0, stream, (const bf16_t*)delta_raw, ld, (const bf16_t*)a_raw, lda,
                       (const bf16_t*)o_raw, ldo, roi, (const bf16_t*)curr, (const bf16_t*)next, workspace, nullptr, nullptr, nullptr, nullptr,
                       CT_curr, CT_next, HW, ps, npatch, clip, shift);
  hipLaunchKernelGGL(sum4_final_kernel, dim3(1), dim3(256), 0, stream, workspace, sums, B * npatch);
  MLA_LAUNCH_CHECK();
}
// coef[3] (device) = { g / n_roi_elems, 0.01 * g / n_bg_elems, -0.1 * g / n_all_elems } (zero where a term is absent);
// outputs: ddelta_raw [B, npatch, ld] bf16 (padding columns zeroed), dalpha_raw [B * npatch] fp32, doff_raw [B * npatch, 2] fp32
extern "C" int mla_imgroi_bwd(const void* delta_raw, int ld, const void* a_raw, int lda, const void* o_raw, int ldo,
                              const unsigned char* roi, const void* curr, const void* next, int img_fp32, const float* coef,
                              void* ddelta_raw, float* dalpha_raw, float* doff_raw, int B, int CT_curr, int CT_next, int HW, int ps,
                              float clip, float shift, hipStream_t stream) {
  MLA_CHECK_ARG(delta_raw && a_raw && o_raw && roi && curr && next && coef && ddelta_raw && dalpha_raw && doff_raw, "mla_imgroi_bwd: null pointer");
  const int npatch = (HW / ps) * (HW / ps);
  if (img_fp32)
    hipLaunchKernelGGL((imgroi_kernel<float, 1>), dim3(B * npatch), dim3(256), 0, stream, (const bf16_t*)delta_raw, ld, (const bf16_t*)a_raw, lda,
                       (const bf16_t*)o_raw, ldo, roi, (const float*)curr, (const float*)next, nullptr, coef, (bf16_t*)ddelta_raw, dalpha_raw,
                       doff_raw, CT_curr, CT_next, HW, ps, npatch, clip, shift);
  else
    hipLaunchKernelGGL((imgroi_kernel<bf16_t, 1>), dim3(B * npatch), dim3(256), 0, stream, (const bf16_t*)delta_raw, ld, (const bf16_t*)a_raw, lda,
                       (const bf16_t*)o_raw, ldo, roi, (const bf16_t*)curr, (const bf16_t*)next, nullptr, coef, (bf16_t*)ddelta_raw, dalpha_raw,
                       doff_raw, CT_curr, CT_next, HW, ps, npatch, clip, shift);
  MLA_LAUNCH_CHECK();
}
