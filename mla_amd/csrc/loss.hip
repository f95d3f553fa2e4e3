// Loss-side kernels: row-wise softmax cross-entropy (lm_head CE, contrastive InfoNCE), L2 row normalisation.
// Reference: CrossEntropyLoss at transformers/models/llama/modeling_llama.py:1258-1269;
//            CoordinateAwareContrastiveLoss.forward models/mla/fuser/contrastive.py:185-215.
#include "common.h"
#include <math.h>

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }

// One block per row. lse[r] = logsumexp(logits[r, :ncols]); loss[r] = lse - logits[r, label] (0 when label is
// ignore_index or out of range).  Online max/sum in one pass.
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, long long ld, const long long* __restrict__ labels,
                                                     float* __restrict__ loss, float* __restrict__ lse, int rows, int ncols,
                                                     long long ignore_index) {
  __shared__ float scratch[16];
  const int r = blockIdx.x;
  const T* row = logits + (long long)r * ld;
  float m = -INFINITY, s = 0.f;
  int j0 = 0;
  if (sizeof(T) == 4 && ((((uintptr_t)row) & 15) == 0)) {
    // fp32 logits, 16-B loads: the running maximum moves once per four columns (branch-free), 5 exponentials per 4 values. The
    // scalar walk below read 4 B per lane and ran at 3.0 TB/s on the [17 536, 32 064] lm_head logits.
    const f32x4_t* row4 = (const f32x4_t*)row;
    const int n4 = ncols >> 2;
    for (int j = threadIdx.x; j < n4; j += 256) {
      const f32x4_t x = row4[j];
      const float mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
      const float mn = fmaxf(m, mx);
      if (mn != -INFINITY) {
        s = s * __expf(m - mn) + ((__expf(x[0] - mn) + __expf(x[1] - mn)) + (__expf(x[2] - mn) + __expf(x[3] - mn)));
        m = mn;
      }
    }
    j0 = n4 << 2;
  }
  for (int j = j0 + threadIdx.x; j < ncols; j += 256) {
    const float x = ldf<T>(row + j);
    if (x > m) { s = s * __expf(m - x) + 1.f; m = x; }
    else s += __expf(x - m);
  }
  const float gm = block_max(m, scratch);
  const float contrib = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
  const float gs = block_sum(contrib, scratch);
  if (threadIdx.x == 0) {
    const float l = gm + logf(gs);
    if (lse) lse[r] = l;
    if (loss) {
      const long long lab = labels ? labels[r] : (long long)r;
      loss[r] = (lab == ignore_index || lab < 0 || lab >= ncols) ? 0.f : (l - ldf<T>(row + lab));
    }
  }
}

// Symmetric InfoNCE gradient on L[Mp, Mp] (fp32, only the leading M x M block is real):
// dL[i][j] = gscale[0] * (exp(L_ij - rlse_i) + exp(L_ij - clse_j) - 2*delta_ij) / (2 M)   for i,j < M, else 0
__global__ __launch_bounds__(256) void infonce_bwd_kernel(const float* __restrict__ L, const float* __restrict__ rlse,
                                                          const float* __restrict__ clse, const float* __restrict__ gscale,
                                                          bf16_t* __restrict__ dL, int M, int Mp) {
  const float gs = gscale[0] / (2.f * (float)M);
  if ((Mp & 3) == 0 && ((((uintptr_t)L) | ((uintptr_t)clse)) & 15) == 0 && (((uintptr_t)dL) & 7) == 0) {   // 4 columns per lane: 16-B / 8-B accesses
    const int q = Mp >> 2;
    const long long total4 = (long long)Mp * q;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long long)gridDim.x * 256) {
      const int i = (int)(idx / q), j = (int)(idx % q) * 4;
      u32x2_t o = {0u, 0u};
      if (i < M && j < M) {
        const f32x4_t x = *(const f32x4_t*)(L + (long long)i * Mp + j);
        const float ri = rlse[i];
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          v[k] = (j + k < M) ? gs * (__expf(x[k] - ri) + __expf(x[k] - clse[j + k]) - (i == j + k ? 2.f : 0.f)) : 0.f;
        o[0] = pack2bf(v[0], v[1]);
        o[1] = pack2bf(v[2], v[3]);
      }
      *(u32x2_t*)(dL + (long long)i * Mp + j) = o;
    }
    return;
  }
  const long long total = (long long)Mp * Mp;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int i = (int)(idx / Mp), j = (int)(idx % Mp);
    float v = 0.f;
    if (i < M && j < M) {
      const float x = L[idx];
      v = gs * (__expf(x - rlse[i]) + __expf(x - clse[j]) - (i == j ? 2.f : 0.f));
    }
    dL[idx] = f2bf(v);
  }
}

// generic CE backward: dlogits[r][j] = g[r] * (softmax_j - [j == label])  (bf16 out), g[r] = 0 for ignored rows
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, long long ld, const long long* __restrict__ labels,
                                                     const float* __restrict__ lse, const float* __restrict__ gscale,
                                                     float inv_count, bf16_t* __restrict__ dlogits, long long ldd, int rows,
                                                     int ncols, long long ignore_index) {
  const int r = blockIdx.x;
  const long long lab = labels[r];
  const bool ign = (lab == ignore_index || lab < 0 || lab >= ncols);
  const float g = ign ? 0.f : gscale[0] * inv_count;
  const float l = lse[r];
  for (int j = threadIdx.x; j < ncols; j += 256) {
    const float pr = __expf(ldf<T>(logits + (long long)r * ld + j) - l);
    dlogits[(long long)r * ldd + j] = f2bf(g * (pr - (j == lab ? 1.f : 0.f)));
  }
}

// y = x / max(||x||_2, eps) per row (F.normalize); one wave per row, ncols % 8 == 0
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                         float* __restrict__ norms, int rows, int ncols, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = ncols >> 3;
  float ss = 0.f;
  for (int c = lane; c < nch; c += 64) {
    const u32x4_t w = *(const u32x4_t*)(x + (long long)row * ncols + c * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) ss += bflo(w[k]) * bflo(w[k]) + bfhi(w[k]) * bfhi(w[k]);
  }
  ss = wave_sum(ss);
  const float nrm = fmaxf(sqrtf(ss), eps);
  if (lane == 0 && norms) norms[row] = nrm;
  const float inv = 1.f / nrm;
  for (int c = lane; c < nch; c += 64) {
    const u32x4_t w = *(const u32x4_t*)(x + (long long)row * ncols + c * 8);
    u32x4_t o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack2bf(bflo(w[k]) * inv, bfhi(w[k]) * inv);
    *(u32x4_t*)(y + (long long)row * ncols + c * 8) = o;
  }
}
// dx = (dy - y * <y, dy>) / norm
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                         const float* __restrict__ norms, bf16_t* __restrict__ dx, int rows,
                                                         int ncols) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = ncols >> 3;
  float dot = 0.f;
  for (int c = lane; c < nch; c += 64) {
    const u32x4_t a = *(const u32x4_t*)(y + (long long)row * ncols + c * 8);
    const u32x4_t b = *(const u32x4_t*)(dy + (long long)row * ncols + c * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k) dot += bflo(a[k]) * bflo(b[k]) + bfhi(a[k]) * bfhi(b[k]);
  }
  dot = wave_sum(dot);
  const float inv = 1.f / norms[row];
  for (int c = lane; c < nch; c += 64) {
    const u32x4_t a = *(const u32x4_t*)(y + (long long)row * ncols + c * 8);
    const u32x4_t b = *(const u32x4_t*)(dy + (long long)row * ncols + c * 8);
    u32x4_t o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = pack2bf((bflo(b[k]) - bflo(a[k]) * dot) * inv, (bfhi(b[k]) - bfhi(a[k]) * dot) * inv);
    *(u32x4_t*)(dx + (long long)row * ncols + c * 8) = o;
  }
}

}  // namespace

extern "C" int mla_ce_fwd(const void* logits, int logits_fp32, long long ld, const long long* labels, float* loss, float* lse,
                          int rows, int ncols, long long ignore_index, hipStream_t stream) {
  MLA_CHECK_ARG(logits && (loss || lse) && rows > 0 && ncols > 0, "mla_ce_fwd: bad args");
  if (logits_fp32)
    hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(rows), dim3(256), 0, stream, (const float*)logits, ld, labels, loss, lse, rows, ncols, ignore_index);
  else
    hipLaunchKernelGGL(ce_fwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, stream, (const bf16_t*)logits, ld, labels, loss, lse, rows, ncols, ignore_index);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_ce_bwd(const void* logits, int logits_fp32, long long ld, const long long* labels, const float* lse,
                          const float* gscale, float inv_count, void* dlogits, long long ldd, int rows, int ncols,
                          long long ignore_index, hipStream_t stream) {
  MLA_CHECK_ARG(logits && labels && lse && gscale && dlogits, "mla_ce_bwd: bad args");
  if (logits_fp32)
    hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(rows), dim3(256), 0, stream, (const float*)logits, ld, labels, lse, gscale, inv_count, (bf16_t*)dlogits, ldd, rows, ncols, ignore_index);
  else
    hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, stream, (const bf16_t*)logits, ld, labels, lse, gscale, inv_count, (bf16_t*)dlogits, ldd, rows, ncols, ignore_index);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_infonce_bwd(const float* L, const float* rlse, const float* clse, const float* gscale, void* dL, int M, int Mp,
                               hipStream_t stream) {
  MLA_CHECK_ARG(L && rlse && clse && gscale && dL && M > 0 && Mp >= M, "mla_infonce_bwd: bad args");
  long long nb = ((long long)Mp * Mp + 255) / 256; if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(infonce_bwd_kernel, dim3((int)nb), dim3(256), 0, stream, L, rlse, clse, gscale, (bf16_t*)dL, M, Mp);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_l2norm_fwd(const void* x, void* y, float* norms, int rows, int ncols, float eps, hipStream_t stream) {
  MLA_CHECK_ARG(x && y && ncols % 8 == 0, "mla_l2norm_fwd: bad args");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, norms, rows, ncols, eps);
  MLA_LAUNCH_CHECK();
}
extern "C" int mla_l2norm_bwd(const void* dy, const void* y, const float* norms, void* dx, int rows, int ncols, hipStream_t stream) {
  MLA_CHECK_ARG(dy && y && norms && dx && ncols % 8 == 0, "mla_l2norm_bwd: bad args");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)y, norms, (bf16_t*)dx, rows, ncols);
  MLA_LAUNCH_CHECK();
}
