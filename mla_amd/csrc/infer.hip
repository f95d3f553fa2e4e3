// Inference-side kernels (round 6; SURVEY 8f rank 2: predict_action_diff, models/mla/model_mla.py:592-775 +
// models/diffusion/gaussian_diffusion.py:608-688). The reference re-runs the WHOLE 548-token forward for each of the 8 DDIM steps
// although everything in front of the [t, x] tokens is identical in all of them. mla_amd/infer.py runs one prefill that keeps the packed
// post-RoPE q|k|v rows of every layer, then 8 passes over the 2 suffix rows per sample. Those passes are weight-streaming
// (13.5 GB of bf16 weights per pass, ~2.5 ms at HBM rate), not MFMA work:
//   mla_gemv_bf16      out[m, n] = sum_k x[m, k] W[n, k] (+ residual), M <= 8 rows: every W row is read once, 16 B per lane, non-temporal,
//                      one wave per row pair, 256 B per lane in flight; x lives in LDS as bf16; fp32 accumulation.
//   mla_attn_decode    R <= 8 new query rows per (sample, head) against the cached keys / values [0, S_kv - R + r]: scores -> LDS,
//                      softmax per query, P V with 4 key slices per block; head_dim 128.
// Both are HBM-bound by construction: algorithmic bytes = the weight matrix (gemv) / the K and V rows of the head (decode).
#include "common.h"

namespace {

constexpr int GEMV_MMAX = 8;
constexpr int GEMV_ROWS = 2;          // W rows per wave
constexpr int GEMV_UNR = 8;           // K steps (64 lanes x 16 B each) whose loads are all issued before the first use

// x: [M, K] bf16 rows (ldx); W: [N, K] bf16, k-contiguous (ldw); out row m lives at out + (m / rpb) * out_bs + (m % rpb) * ldo
// (rpb rows per sample: lets the q|k|v rows of the suffix land directly in the per-sample cache slots); residual addressed like x
// with ld_res. K % 8 == 0, 16-B aligned rows.
// A 7B projection is only 33-90 MB: the kernel lives for a few microseconds and what it reaches is decided by the bytes in flight
// (Little's law: ~8 TB/s x ~2 us of loaded latency = 16 MB chip-wide), not by a loop. Lane map: 64 lanes per W row (a wave reads 1 KiB
// of a row per instruction), GEMV_ROWS rows per wave, the loads of GEMV_UNR K steps issued back to back into registers
// (2 x 8 x 16 B = 256 B per lane in flight) before the first FMA, one wave per row pair so that N = 4096 already puts 2 048 waves on the
// chip. Measured on the 7B suffix pass (12.95 GB of weights): 64 lanes per row with 4 rows per wave and no explicit batching 7.3 ms;
// 16 lanes per row / 8 rows per wave (four times fewer waves) 10.4 ms; this form 4.9 ms, 10-17 us per 33 MB projection (requesting the
// first row pair's weights BEFORE the input staging, to hide the staging / fused-RMSNorm prologue, cost 224 registers and was slower:
// 22.9 vs 16.8 us); profiles/r6_infer_latency.txt.
// PRE: what happens to the input rows on their way into LDS (every workgroup does it for itself -- M x K elements, a few KB -- instead
// of a separate launch in front of every projection: a pass over the suffix rows is launch-gap-bound once the weights stream at HBM rate)
//   0  nothing          1  LlamaRMSNorm (modeling_llama.py:76-90, same arithmetic and cast order as rmsnorm_fwd_kernel)
//   2  SwiGLU: x is the packed gate|up row [2 K], the GEMV input is silu(gate) * up (swiglu_fwd_elem, common.h)
__device__ __forceinline__ void unpack8f(const u32x4_t v, float* f) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = bflo(v[j]); f[2 * j + 1] = bfhi(v[j]); }
}
__device__ __forceinline__ u32x4_t pack8f(const float* f) {
  u32x4_t v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = pack2bf(f[2 * j], f[2 * j + 1]);
  return v;
}
template <int M, int PRE>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ x, long long ldx, const bf16_t* __restrict__ W, long long ldw,
                                                   bf16_t* __restrict__ out, long long ldo, long long out_bs, int rpb,
                                                   const bf16_t* __restrict__ res, long long ld_res, int N, int K,
                                                   const bf16_t* __restrict__ pre_w, float eps, const float* __restrict__ rope_cos,
                                                   const float* __restrict__ rope_sin, int rope_cols) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int kc = K >> 3;
  u32x4_t* xs = (u32x4_t*)smem;                       // [M][K / 8] chunks of 8 bf16
  float* scratch = (float*)(smem + (size_t)M * K * 2);   // 16 floats behind the rows (block reductions of the fused RMSNorm)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sets = (N + GEMV_ROWS - 1) / GEMV_ROWS;
  const int steps = (kc + 63) >> 6;                   // K steps of 64 chunks
  // input rows -> LDS, four independent 16-B loads per thread in flight (one load per loop trip cost a dependent round trip each:
  // ~6 us in front of every projection)
  for (int i0 = threadIdx.x; i0 < M * kc; i0 += 256 * 4) {
    u32x4_t a[4], b2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = i0 + t * 256;
      const int ii = i < M * kc ? i : M * kc - 1;
      const int m = ii / kc, c = ii - m * kc;
      a[t] = *(const u32x4_t*)(x + (long long)m * ldx + c * 8);
      if (PRE == 2) b2[t] = *(const u32x4_t*)(x + (long long)m * ldx + K + c * 8);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = i0 + t * 256;
      if (i < M * kc) {
        if (PRE == 2) {
          float g[8], u[8], o[8];
          unpack8f(a[t], g);
          unpack8f(b2[t], u);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = swiglu_fwd_elem(g[j], u[j]);
          xs[i] = pack8f(o);
        } else {
          xs[i] = a[t];
        }
      }
    }
  }
  __syncthreads();
  if (PRE == 1) {
    float rstd[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float ss = 0.f;
      for (int c = threadIdx.x; c < kc; c += 256) {
        float f[8];
        unpack8f(xs[m * kc + c], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
      }
      ss = block_sum(ss, scratch);
      rstd[m] = 1.0f / sqrtf(ss / (float)K + eps);
    }
    for (int c = threadIdx.x; c < kc; c += 256) {
      float wv[8];
      unpack8f(*(const u32x4_t*)(pre_w + c * 8), wv);
#pragma unroll
      for (int m = 0; m < M; ++m) {
        float f[8], o[8];
        unpack8f(xs[m * kc + c], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = wv[j] * bf2f(f2bf(f[j] * rstd[m]));
        xs[m * kc + c] = pack8f(o);
      }
    }
    __syncthreads();
  }
  for (int set = blockIdx.x * 4 + wave; set < sets; set += gridDim.x * 4) {
    // rows of this wave: a consecutive pair -- or, in the rotary columns [0, rope_cols) of a fused q|k|v projection, channel d of a head
    // and its rotation partner d + 64 (apply_rotary_pos_emb, modeling_llama.py:184-208), so that the epilogue can rotate them
    const bool rot = rope_cos != nullptr && set * GEMV_ROWS < rope_cols;
    const int n0 = rot ? (set >> 6) * 128 + (set & 63) : set * GEMV_ROWS;
    const int nstep = rot ? 64 : 1;
    float acc[GEMV_ROWS][M];
    const bf16_t* wr[GEMV_ROWS];
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r) {
      wr[r] = W + (long long)(n0 + r * nstep < N ? n0 + r * nstep : N - 1) * ldw;
#pragma unroll
      for (int m = 0; m < M; ++m) acc[r][m] = 0.f;
    }
    for (int s0 = 0; s0 < steps; s0 += GEMV_UNR) {
      u32x4_t w[GEMV_UNR][GEMV_ROWS];
#pragma unroll
      for (int u = 0; u < GEMV_UNR; ++u) {
        const int c = (s0 + u) * 64 + lane;
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r)
          w[u][r] = c < kc ? __builtin_nontemporal_load((const u32x4_t*)(wr[r] + c * 8)) : u32x4_t{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < GEMV_UNR; ++u) {
        const int c = (s0 + u) * 64 + lane;
        if ((s0 + u) * 64 < kc) {                     // (wave-uniform: whole steps beyond K are skipped)
          const int cc = c < kc ? c : kc - 1;          // lanes past the end read a valid chunk against their zero weights
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const u32x4_t xv = xs[m * kc + cc];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float xl = bflo(xv[j]), xh = bfhi(xv[j]);
#pragma unroll
              for (int r = 0; r < GEMV_ROWS; ++r) acc[r][m] = fmaf(bfhi(w[u][r][j]), xh, fmaf(bflo(w[u][r][j]), xl, acc[r][m]));
            }
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r)
#pragma unroll
      for (int m = 0; m < M; ++m) acc[r][m] = wave_sum(acc[r][m]);
    if (lane < GEMV_ROWS * M) {
      const int r = lane / M, m = lane - r * M;
      if (n0 + r * nstep < N) {
        float v = 0.f, partner = 0.f;
#pragma unroll
        for (int rr = 0; rr < GEMV_ROWS; ++rr)
#pragma unroll
          for (int mm = 0; mm < M; ++mm) {
            v = (rr == r && mm == m) ? acc[rr][mm] : v;
            partner = (rr != r && mm == m) ? acc[rr][mm] : partner;
          }
        if (res) v += bf2f(res[(long long)m * ld_res + n0 + r * nstep]);
        if (rot) {
          // the arithmetic of rope_kernel (elementwise.hip) on the bf16-rounded projection: a' = a cos - b sin, b' = b cos + a sin
          const int d = n0 & 63, pos = m % rpb;                    // table row = the row's index inside its sample's block of new rows
          const float c = rope_cos[pos * 64 + d], sn = rope_sin[pos * 64 + d];
          const float me = bf2f(f2bf(v)), other = bf2f(f2bf(partner));
          v = r == 0 ? fmaf(me, c, -(other * sn)) : fmaf(me, c, other * sn);
        }
        out[(long long)(m / rpb) * out_bs + (long long)(m % rpb) * ldo + n0 + r * nstep] = f2bf(v);
      }
    }
  }
}

// One block per (sample, head): 4 waves. q rows: q + (b * bs + (S_kv - R + r) * ld) + h * 128 (the new rows are the LAST R rows of the
// cache); query r sees keys [0, S_kv - R + r]. scale applied to the scores; softmax in fp32; o: [B * R, H * 128] bf16.
// Both passes over the keys map 16 lanes to one key row (8 channels = 16 B per lane) and 16 keys to one block step, and issue the
// loads of DEC_U steps before the first use: the first version (one dependent 16-B load per step in the score loop, one 4-B load per
// key in the P V loop) took 110 us per call at S_kv = 547 -- 35 + 138 serial round trips -- against ~15 us of the head's 281 KB at the
// rate one CU streams.
constexpr int DEC_RMAX = 8;
constexpr int DEC_U = 6;
constexpr int DEC_NW = 8;                              // waves per block: 32 keys per block step
__global__ __launch_bounds__(64 * DEC_NW) void attn_decode_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                                  bf16_t* __restrict__ o, int H, int S_kv, int R, long long ld, long long bs,
                                                                  long long ld_o, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc = (float*)smem;                            // [R][S_kv] scores -> probabilities
  float* part = sc + (size_t)R * S_kv;                 // [DEC_NW waves][R][128] partial outputs
  const int h = blockIdx.x % H, b = blockIdx.x / H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bf16_t* kb = k + b * bs + h * 128;
  const bf16_t* vb = v + b * bs + h * 128;
  const bf16_t* qb = q + b * bs + (long long)(S_kv - R) * ld + h * 128;
  const int sub = lane & 15, kg = wave * 4 + (lane >> 4);
  constexpr int KSTEP = 4 * DEC_NW;
  // ---- scores
  float qf[DEC_RMAX][8];
#pragma unroll
  for (int r = 0; r < DEC_RMAX; ++r) {
    if (r < R) {
      const u32x4_t qv = *(const u32x4_t*)(qb + (long long)r * ld + sub * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) { qf[r][2 * j] = bflo(qv[j]); qf[r][2 * j + 1] = bfhi(qv[j]); }
    }
  }
  for (int j0 = 0; j0 < S_kv; j0 += KSTEP * DEC_U) {
    u32x4_t kv[DEC_U];
#pragma unroll
    for (int u = 0; u < DEC_U; ++u) {
      const int j = j0 + u * KSTEP + kg;
      kv[u] = *(const u32x4_t*)(kb + (long long)(j < S_kv ? j : S_kv - 1) * ld + sub * 8);
    }
#pragma unroll
    for (int u = 0; u < DEC_U; ++u) {
      const int j = j0 + u * KSTEP + kg;
      float kf[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { kf[2 * e] = bflo(kv[u][e]); kf[2 * e + 1] = bfhi(kv[u][e]); }
#pragma unroll
      for (int r = 0; r < DEC_RMAX; ++r) {
        if (r < R) {
          float s = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) s = fmaf(qf[r][e], kf[e], s);
#pragma unroll
          for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
          if (sub == 0 && j < S_kv) sc[r * S_kv + j] = (j <= S_kv - R + r) ? s * scale : -INFINITY;
        }
      }
    }
  }
  // the first batch of V rows is requested before the softmax: it depends on nothing computed here
  u32x4_t vv0[DEC_U];
#pragma unroll
  for (int u = 0; u < DEC_U; ++u) {
    const int j = u * KSTEP + kg;
    vv0[u] = *(const u32x4_t*)(vb + (long long)(j < S_kv ? j : S_kv - 1) * ld + sub * 8);
  }
  __syncthreads();
  // ---- softmax: wave w normalises queries w, w + DEC_NW
  for (int r = wave; r < R; r += DEC_NW) {
    float m = -INFINITY;
    for (int j = lane; j < S_kv; j += 64) m = fmaxf(m, sc[r * S_kv + j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < S_kv; j += 64) { const float e = __expf(sc[r * S_kv + j] - m); sc[r * S_kv + j] = e; sum += e; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < S_kv; j += 64) sc[r * S_kv + j] = bf2f(f2bf(sc[r * S_kv + j] * inv));   // P feeds the P V product as bf16, like the flash kernel's
  }
  __syncthreads();
  // ---- P V: lane owns channels sub * 8 .. + 7 of the keys of its group
  float acc[DEC_RMAX][8];
#pragma unroll
  for (int r = 0; r < DEC_RMAX; ++r)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[r][e] = 0.f;
  for (int j0 = 0; j0 < S_kv; j0 += KSTEP * DEC_U) {
    u32x4_t vv[DEC_U];
    if (j0 == 0) {
#pragma unroll
      for (int u = 0; u < DEC_U; ++u) vv[u] = vv0[u];
    } else {
#pragma unroll
      for (int u = 0; u < DEC_U; ++u) {
        const int j = j0 + u * KSTEP + kg;
        vv[u] = *(const u32x4_t*)(vb + (long long)(j < S_kv ? j : S_kv - 1) * ld + sub * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < DEC_U; ++u) {
      const int j = j0 + u * KSTEP + kg;
      if (j < S_kv) {
        float vf[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[2 * e] = bflo(vv[u][e]); vf[2 * e + 1] = bfhi(vv[u][e]); }
#pragma unroll
        for (int r = 0; r < DEC_RMAX; ++r)
          if (r < R) {
            const float p = sc[r * S_kv + j];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[r][e] = fmaf(p, vf[e], acc[r][e]);
          }
      }
    }
  }
  // the four key groups of a wave first (lanes l, l ^ 16, l ^ 32, l ^ 48), then the waves through LDS
#pragma unroll
  for (int r = 0; r < DEC_RMAX; ++r)
    if (r < R) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float a = acc[r][e];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (lane < 16) part[(wave * R + r) * 128 + sub * 8 + e] = a;
      }
    }
  __syncthreads();
  for (int i = threadIdx.x; i < R * 128; i += 64 * DEC_NW) {
    const int r = i >> 7, c = i & 127;
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < DEC_NW; ++g) s += part[(g * R + r) * 128 + c];
    o[(long long)(b * R + r) * ld_o + h * 128 + c] = f2bf(s);
  }
}

}  // namespace

#define AL16(p) ((((uintptr_t)(p)) & 15) == 0)

extern "C" int mla_gemv_bf16(const void* x, long long ldx, const void* W, long long ldw, void* out, long long ldo, long long out_batch_stride,
                             int rows_per_batch, const void* residual, long long ld_res, int M, int N, int K, int pre, const void* pre_w, float eps,
                             const float* rope_cos, const float* rope_sin, int rope_cols, hipStream_t stream) {
  MLA_CHECK_ARG(x && W && out, "mla_gemv_bf16: null pointer");
  MLA_CHECK_ARG((rope_cos == nullptr) == (rope_sin == nullptr) && (!rope_cos || (rope_cols > 0 && rope_cols % 128 == 0 && rope_cols <= N && !residual)),
                "mla_gemv_bf16: the RoPE epilogue needs both tables, rope_cols a multiple of 128 and <= N, and no residual");
  MLA_CHECK_ARG(pre >= 0 && pre <= 2 && (pre != 1 || (pre_w && AL16(pre_w))), "mla_gemv_bf16: pre must be 0, 1 (RMSNorm: 16-B aligned weight needed) or 2 (SwiGLU)");
  MLA_CHECK_ARG(M >= 1 && M <= GEMV_MMAX && N >= 1 && K >= 8 && K % 8 == 0 && rows_per_batch >= 1, "mla_gemv_bf16: 1 <= M <= 8, K %% 8 == 0 required (M %d, N %d, K %d)", M, N, K);
  MLA_CHECK_ARG(AL16(x) && AL16(W) && ldx % 8 == 0 && ldw % 8 == 0, "mla_gemv_bf16: x / W rows must be 16-B aligned");
  const size_t lds = (size_t)M * K * 2 + 64;
  MLA_CHECK_ARG(lds <= 160 * 1024, "mla_gemv_bf16: M x K x 2 bytes of input rows (+ 64) must fit the 160 KiB of LDS (M %d, K %d)", M, K);
  static int cus = 0;
  if (!cus) { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); cus = (hipGetDeviceProperties(&p, d) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
  const int sets = (N + GEMV_ROWS - 1) / GEMV_ROWS;
  int blocks = (sets + 3) / 4;
  int per_cu = (int)((160 * 1024) / (lds > 20 * 1024 ? lds : 20 * 1024));      // resident workgroups per CU: LDS for x, at most 32 waves
  per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
  const int cap = cus * per_cu;
  if (blocks > cap) blocks = cap;
#define MLA_GEMV_LAUNCH(MM, PP)                                                                                                        \
  {                                                                                                                                    \
    static bool attr = false;                                                                                                          \
    if (!attr) { (void)hipFuncSetAttribute((const void*)gemv_kernel<MM, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
    hipLaunchKernelGGL((gemv_kernel<MM, PP>), dim3(blocks), dim3(256), lds, stream, (const bf16_t*)x, ldx, (const bf16_t*)W, ldw, (bf16_t*)out, ldo, \
                       out_batch_stride, rows_per_batch, (const bf16_t*)residual, ld_res, N, K, (const bf16_t*)pre_w, eps, rope_cos, rope_sin, \
                       rope_cos ? rope_cols : 0);                                                                                      \
  }
#define MLA_GEMV_CASE(MM)                                                                                                              \
  case MM:                                                                                                                             \
    if (pre == 0) MLA_GEMV_LAUNCH(MM, 0) else if (pre == 1) MLA_GEMV_LAUNCH(MM, 1) else MLA_GEMV_LAUNCH(MM, 2)                          \
    break;
  switch (M) {
    MLA_GEMV_CASE(1) MLA_GEMV_CASE(2) MLA_GEMV_CASE(3) MLA_GEMV_CASE(4) MLA_GEMV_CASE(5) MLA_GEMV_CASE(6) MLA_GEMV_CASE(7) MLA_GEMV_CASE(8)
  }
#undef MLA_GEMV_CASE
#undef MLA_GEMV_LAUNCH
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_attn_decode(const void* q, const void* k, const void* v, void* o, int B, int H, int head_dim, int S_kv, int R, long long ld,
                               long long batch_stride, long long ld_o, float scale, hipStream_t stream) {
  MLA_CHECK_ARG(q && k && v && o, "mla_attn_decode: null pointer");
  MLA_CHECK_ARG(head_dim == 128, "mla_attn_decode: head_dim must be 128 (got %d)", head_dim);
  MLA_CHECK_ARG(B >= 1 && H >= 1 && R >= 1 && R <= DEC_RMAX && S_kv >= R, "mla_attn_decode: 1 <= R <= 8 <= S_kv required (R %d, S_kv %d)", R, S_kv);
  MLA_CHECK_ARG(AL16(q) && AL16(k) && AL16(v) && ld % 8 == 0 && batch_stride % 8 == 0 && ld_o % 2 == 0, "mla_attn_decode: 16-B aligned rows required");
  const size_t lds = ((size_t)R * S_kv + DEC_NW * (size_t)R * 128) * 4;
  MLA_CHECK_ARG(lds <= 160 * 1024, "mla_attn_decode: R x S_kv scores do not fit LDS (R %d, S_kv %d)", R, S_kv);
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)attn_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(attn_decode_kernel, dim3(B * H), dim3(64 * DEC_NW), lds, stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, H, S_kv,
                     R, ld, batch_stride, ld_o, scale);
  MLA_LAUNCH_CHECK();
}
