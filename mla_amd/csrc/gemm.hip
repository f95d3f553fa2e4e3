// bf16 MFMA GEMM for gfx950 (MI355X): C[M,N] = alpha * sum_k Aop(m,k) * Bop(n,k) (+ bias[n]) (+ R[m,n]) (+ C_old)
//
// One kernel family covers the three products a Linear layer needs
// (reference: transformers/models/llama/modeling_llama.py:211-242, 320-402 -- every q/k/v/o/gate/up/down
//  projection is an nn.Linear; autograd derives the two backward products):
//   forward  y  = x W^T      : A = x  [M,K]  k-contiguous (mode 0), B = W  [N,K] k-contiguous (mode 0)
//   dgrad    dx = dy W       : A = dy [M,K'] k-contiguous (mode 0), B = W  [K',N] reduction-major (mode 1)
//   wgrad    dW = dy^T x     : A = dy [K',M] reduction-major (mode 1), B = x [K',N] reduction-major (mode 1)
// so no transposed weight/activation copies ever exist in HBM.
//
// Structure (DESIGN.md "GEMM"): 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave 64x64 via 4x4
// v_mfma_f32_16x16x32_bf16 fragments; operands staged HBM->LDS with global_load_lds_dwordx4 (no VGPR round trip),
// double-buffered (64 KiB LDS, 2 blocks/CU); LDS images are lane-linear with the XOR swizzle applied on the
// *source* address (k-contiguous tiles: 16-B chunk ^= row&7, read with ds_read_b128; reduction-major tiles:
// chunk ^= h(krow)<<1, read with ds_read_b64_tr_b16). MFMA operands are passed swapped (B-fragment as the A
// operand) so each lane ends up with 4 consecutive output columns of one row -> 8/16-byte stores.
// blockIdx is remapped so every XCD (private L2) walks a contiguous, M-grouped range of tiles.
#include "common.h"

#include "gemm_args.h"

int mla_gemm256_dispatch(const void* args, int a_mode, int b_mode, size_t ws_bytes, hipStream_t stream, int* sq_slots = nullptr);  // gemm256.hip
extern "C" int mla_gemm_sq_slots(int M, int N, int K, size_t workspace_bytes);                                                          // gemm256.hip
#ifdef MLA_EXPERIMENTAL_KERNELS   // opt-in experiment kernels (build.sh: MLA_EXPERIMENTAL=1); the product library carries only kernels on the path
int mla_gemm_asm_dispatch(const void* args, hipStream_t stream);
int mla_gemm_asm4_dispatch(const void* args, hipStream_t stream);                           // gemm_asm.hip
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 16384;  // 128x64 bf16

__device__ __forceinline__ int hsw(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Stage one operand tile into LDS. MODE 0: global is [rows][K] (ld), tile = rows i0..i0+127 x k0..k0+63.
// MODE 1: global is [K][cols] (ld), tile = k0..k0+63 x cols i0..i0+127.  Out-of-range rows/cols/k are clamped to
// valid addresses: clamped row/col data only reaches masked outputs, clamped k data is never multiplied (the
// K tail is a multiple of 32 and the second k-step is skipped).
// UNTRACKED: the copy is issued from inline assembly (glds16_untracked): kernels that read fragments with ds_read_b64_tr_b16 must
// not let the compiler see the LDS-DMA copies, or it drains the prefetch in front of every such read (common.h).
template <int MODE, bool UNTRACKED>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int ld, int i0, int ilim, int k0, int K,
                                           char* tile, int wave, int lane) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int instr = wave * 4 + it;
    const int p = instr * 64 + lane;
    const bf16_t* src;
    if (MODE == 0) {
      const int row = p >> 3, cp = p & 7;
      const int c = cp ^ (row & 7);
      int gi = i0 + row;
      gi = gi < ilim ? gi : ilim - 1;
      int gk = k0 + c * 8;
      gk = gk < K ? gk : k0;
      src = g + (size_t)gi * ld + gk;
    } else {
      const int kr = p >> 4, cp = p & 15;
      const int c = cp ^ (hsw(kr) << 1);
      int gk = k0 + kr;
      gk = gk < K ? gk : k0;
      int gi = i0 + c * 8;
      gi = (gi + 8 <= ilim) ? gi : 0;
      src = g + (size_t)gk * ld + gi;
    }
    if (UNTRACKED) glds16_untracked(src, tile + instr * 1024);
    else glds16(src, tile + instr * 1024);
  }
}

template <int MODE>
__device__ __forceinline__ bf16x8_t load_frag(const char* tile, int rb, int ks, int lane) {
  const int i = lane & 15, g = lane >> 4;
  if (MODE == 0) {
    const int row = rb * 16 + i;
    const int cp = (ks * 4 + g) ^ (row & 7);
    return *(const bf16x8_t*)(tile + (row * 8 + cp) * 16);
  } else {
    union {
      bf16x8_t v;
      short4_t h[2];
    } u;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int kr = ks * 32 + g * 8 + jj * 4 + (i >> 2);
      const int cp = (rb * 2 + ((i & 3) >> 1)) ^ (hsw(kr) << 1);
      u.h[jj] = lds_tr16_b64(tile + (kr * 16 + cp) * 16 + (i & 1) * 8);
    }
    return u.v;
  }
}

template <int AMODE, int BMODE>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  if (p.n_inner > 0) {   // batched launch: shift the operand base pointers of this z-slice
    const int zo = blockIdx.y / p.n_inner, zi = blockIdx.y % p.n_inner;
    p.A += zo * p.sAo + zi * p.sAi;
    p.B += zo * p.sBo + zi * p.sBi;
    const long long co = zo * p.sCo + zi * p.sCi;
    p.C = p.out_fp32 ? (void*)((float*)p.C + co) : (void*)((bf16_t*)p.C + co);
  }

  const int num_m = (p.M + BM - 1) / BM, num_n = (p.N + BN - 1) / BN;
  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  constexpr int GROUP_M = 8;
  const int in_group = GROUP_M * num_n;
  const int group_id = pid / in_group;
  const int first_m = group_id * GROUP_M;
  const int gsz = (num_m - first_m) < GROUP_M ? (num_m - first_m) : GROUP_M;
  const int pid_m = first_m + (pid % in_group) % gsz;
  const int pid_n = (pid % in_group) / gsz;
  const int m0 = pid_m * BM, n0 = pid_n * BN;

  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nt = (p.K + BK - 1) / BK;
  stage_tile<AMODE, (AMODE != 0 || BMODE != 0)>(p.A, p.lda, m0, p.M, 0, p.K, smem, wave, lane);
  stage_tile<BMODE, (AMODE != 0 || BMODE != 0)>(p.B, p.ldb, n0, p.N, 0, p.K, smem + TILE_BYTES, wave, lane);

  for (int t = 0; t < nt; ++t) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) as a builtin, so that the compiler's wait-count bookkeeping sees it
    asm volatile("" ::: "memory");
    __syncthreads();
    char* cur = smem + (t & 1) * 2 * TILE_BYTES;
    if (t + 1 < nt) {
      char* nxt = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
      stage_tile<AMODE, (AMODE != 0 || BMODE != 0)>(p.A, p.lda, m0, p.M, (t + 1) * BK, p.K, nxt, wave, lane);
      stage_tile<BMODE, (AMODE != 0 || BMODE != 0)>(p.B, p.ldb, n0, p.N, (t + 1) * BK, p.K, nxt + TILE_BYTES, wave, lane);
    }
    const int ksteps = (p.K - t * BK) >= BK ? 2 : 1;
    for (int ks = 0; ks < ksteps; ++ks) {
      bf16x8_t af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = load_frag<AMODE>(cur, wm * 4 + i, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) bfr[i] = load_frag<BMODE>(cur + TILE_BYTES, wn * 4 + i, ks, lane);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
  }

  // epilogue: lane holds C[m][n..n+3]
  const bool vec_ok = ((p.ldc & 3) == 0) && (p.R == nullptr || (p.ldr & 3) == 0);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + (lane >> 4) * 4;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[mi][ni][r] * p.alpha;
      if (n + 3 < p.N && vec_ok) {
        if (p.bias) {
          const u32x2_t bb = *(const u32x2_t*)(p.bias + n);
          v[0] += bflo(bb[0]); v[1] += bfhi(bb[0]); v[2] += bflo(bb[1]); v[3] += bfhi(bb[1]);
        }
        if (p.R) {
          const u32x2_t rr = *(const u32x2_t*)(p.R + (size_t)m * p.ldr + n);
          v[0] += bflo(rr[0]); v[1] += bfhi(rr[0]); v[2] += bflo(rr[1]); v[3] += bfhi(rr[1]);
        }
        if (p.out_fp32) {
          float* c = (float*)p.C + (size_t)m * p.ldc + n;
          f32x4_t o = {v[0], v[1], v[2], v[3]};
          if (p.accumulate) {
            const f32x4_t old = *(const f32x4_t*)c;
            o += old;
          }
          *(f32x4_t*)c = o;
        } else {
          u32x2_t o;
          o[0] = pack2bf(v[0], v[1]);
          o[1] = pack2bf(v[2], v[3]);
          *(u32x2_t*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
        }
      } else {
        for (int r = 0; r < 4 && n + r < p.N; ++r) {
          float x = v[r];
          if (p.bias) x += bf2f(p.bias[n + r]);
          if (p.R) x += bf2f(p.R[(size_t)m * p.ldr + n + r]);
          if (p.out_fp32) {
            float* c = (float*)p.C + (size_t)m * p.ldc + n + r;
            *c = p.accumulate ? (*c + x) : x;
          } else {
            ((bf16_t*)p.C)[(size_t)m * p.ldc + n + r] = f2bf(x);
          }
        }
      }
    }
  }
}

// Generic fallback for shapes the MFMA kernel cannot stage (K % 32 != 0, unaligned leading dimensions: the 7-wide
// action/proprio embedders and the 7-wide FinalLayer output, models/diffusion/models.py:112-123,173-189).
// 64x64 tile, 256 threads, 4x4 outputs per thread, fp32 accumulate, same epilogue semantics.
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmArgs p, int amode, int bmode) {
  __shared__ float As[16][65];
  __shared__ float Bs[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  if (p.n_inner > 0) {
    const int zo = blockIdx.z / p.n_inner, zi = blockIdx.z % p.n_inner;
    p.A += zo * p.sAo + zi * p.sAi;
    p.B += zo * p.sBo + zi * p.sBi;
    const long long co = zo * p.sCo + zi * p.sCi;
    p.C = p.out_fp32 ? (void*)((float*)p.C + co) : (void*)((bf16_t*)p.C + co);
  }
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < p.K; k0 += 16) {
    for (int e = threadIdx.x; e < 64 * 16; e += 256) {
      int i, k;
      if (amode == 0) { i = e >> 4; k = e & 15; } else { k = e >> 6; i = e & 63; }
      const int gm = m0 + i, gk = k0 + k;
      float v = 0.f;
      if (gm < p.M && gk < p.K) v = bf2f(amode == 0 ? p.A[(size_t)gm * p.lda + gk] : p.A[(size_t)gk * p.lda + gm]);
      As[k][i] = v;
      if (bmode == 0) { i = e >> 4; k = e & 15; } else { k = e >> 6; i = e & 63; }
      const int gn = n0 + i;
      const int gk2 = k0 + k;
      v = 0.f;
      if (gn < p.N && gk2 < p.K) v = bf2f(bmode == 0 ? p.B[(size_t)gn * p.ldb + gk2] : p.B[(size_t)gk2 * p.ldb + gn]);
      Bs[k][i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= p.N) continue;
      float x = acc[i][j] * p.alpha;
      if (p.bias) x += bf2f(p.bias[n]);
      if (p.R) x += bf2f(p.R[(size_t)m * p.ldr + n]);
      if (p.out_fp32) {
        float* c = (float*)p.C + (size_t)m * p.ldc + n;
        *c = p.accumulate ? (*c + x) : x;
      } else {
        ((bf16_t*)p.C)[(size_t)m * p.ldc + n] = f2bf(x);
      }
    }
  }
}

template <int AM, int BM_>
int launch128(const GemmArgs& p, hipStream_t stream, int nbatch = 1) {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm128_kernel<AM, BM_>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    attr_set = true;
  }
  const int num_m = (p.M + BM - 1) / BM, num_n = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm128_kernel<AM, BM_>), dim3(num_m * num_n, nbatch), dim3(256), 4 * TILE_BYTES, stream, p);
  MLA_LAUNCH_CHECK();
}

}  // namespace

static int gemm_bf16_impl(const void* A, const void* B, void* C, const void* R, const void* bias, int M, int N,
                          int K, int lda, int ldb, int ldc, int ldr, int a_mode, int b_mode, int out_fp32,
                          int accumulate, float alpha, int force_generic, float* workspace, size_t workspace_bytes,
                          hipStream_t stream, float* sq_out = nullptr, int sq_capacity = 0, int* sq_slots = nullptr) {
  MLA_CHECK_ARG(A && B && C, "mla_gemm_bf16: null operand");
  MLA_CHECK_ARG(M > 0 && N > 0 && K > 0, "mla_gemm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
  MLA_CHECK_ARG((a_mode == 0 || a_mode == 1) && (b_mode == 0 || b_mode == 1), "mla_gemm_bf16: bad modes");
  MLA_CHECK_ARG(lda >= (a_mode == 0 ? K : M) && ldb >= (b_mode == 0 ? K : N) && ldc >= N, "mla_gemm_bf16: bad ld");
  MLA_CHECK_ARG(!accumulate || out_fp32, "mla_gemm_bf16: accumulate needs fp32 output");
  MLA_CHECK_ARG(R == nullptr || ldr >= N, "mla_gemm_bf16: bad ldr");
  GemmArgs p{(const bf16_t*)A, (const bf16_t*)B, C, (const bf16_t*)R, (const bf16_t*)bias, M, N, K,
             lda, ldb, ldc, ldr, out_fp32, accumulate, alpha, force_generic >> 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, workspace};
  force_generic &= 15;
  bool mfma_ok = (force_generic != 1) && (K % 32 == 0) && (lda % 8 == 0) && (ldb % 8 == 0) &&
                 (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0);
  if (a_mode == 1 && (M % 8 != 0)) mfma_ok = false;
  if (b_mode == 1 && (N % 8 != 0)) mfma_ok = false;
  if (out_fp32 && (((uintptr_t)C & 15) != 0)) mfma_ok = false;
  if (!out_fp32 && (((uintptr_t)C & 7) != 0)) mfma_ok = false;
  if (R && (((uintptr_t)R & 7) != 0)) mfma_ok = false;
  if (bias && (((uintptr_t)bias & 7) != 0)) mfma_ok = false;
  // the 256x256 kernel takes every operand layout (round 2: its reduction-major path stages untracked and runs at 1.1-1.3 PFLOP/s,
  // gemm128's at 0.87-0.96; round 1 measured 0.47-0.70 because the compiler drained the ring in front of every ds_read_b64_tr_b16);
  // force_generic == 2 keeps it off, == 3 forces it without the split-K tail (tests / experiments)
#ifdef MLA_EXPERIMENTAL_KERNELS
  // force_generic == 4: the assembly-scheduled 256x256x32 kernel (gemm_asm.hip); K % 128 == 0, N % 4 == 0
  if (mfma_ok && M >= 256 && N >= 256 && (K % 128) == 0 && (N % 4) == 0 && a_mode == 0 && b_mode == 0 && force_generic == 4)
    return mla_gemm_asm_dispatch(&p, stream);
  // force_generic == 5: the 4-wave (128 x 128 per wave) assembly kernel, same eligibility
  if (mfma_ok && M >= 256 && N >= 256 && (K % 128) == 0 && (N % 4) == 0 && a_mode == 0 && b_mode == 0 && force_generic == 5)
    return mla_gemm_asm4_dispatch(&p, stream);
#else
  MLA_CHECK_ARG(force_generic != 4 && force_generic != 5, "mla_gemm_bf16: the assembly experiment kernels are not in this build (MLA_EXPERIMENTAL=1)");
#endif
  if (sq_out) {   // sum-of-squares partials exist in the 256x256 kernel only; the caller sizes the buffer for the worst case
    const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
    MLA_CHECK_ARG(mfma_ok && M >= 256 && N >= 256 && (K % 64) == 0 && force_generic == 0 && sq_slots != nullptr,
                  "mla_gemm_bf16_ws_sq: shape outside the 256x256 kernel");
    const int need = mla_gemm_sq_slots(M, N, K, workspace ? workspace_bytes : 0);
    MLA_CHECK_ARG(need > 0 && sq_capacity >= need, "mla_gemm_bf16_ws_sq: sq_capacity %d < %d (mla_gemm_sq_slots)", sq_capacity, need);
    (void)tiles;
    p.sq_out = sq_out;
    return mla_gemm256_dispatch(&p, a_mode, b_mode, workspace_bytes, stream, sq_slots);
  }
  if (mfma_ok && M >= 256 && N >= 256 && (K % 64) == 0 && (force_generic == 0 || force_generic == 3))
    return mla_gemm256_dispatch(&p, a_mode, b_mode, force_generic == 0 ? workspace_bytes : 0, stream);
  if (mfma_ok) {
    if (a_mode == 0 && b_mode == 0) return launch128<0, 0>(p, stream);
    if (a_mode == 0 && b_mode == 1) return launch128<0, 1>(p, stream);
    if (a_mode == 1 && b_mode == 0) return launch128<1, 0>(p, stream);
    return launch128<1, 1>(p, stream);
  }
  dim3 grid((N + 63) / 64, (M + 63) / 64);
  hipLaunchKernelGGL(gemm_generic_kernel, grid, dim3(256), 0, stream, p, a_mode, b_mode);
  MLA_LAUNCH_CHECK();
}

extern "C" int mla_gemm_bf16(const void* A, const void* B, void* C, const void* R, const void* bias, int M, int N,
                             int K, int lda, int ldb, int ldc, int ldr, int a_mode, int b_mode, int out_fp32,
                             int accumulate, float alpha, int force_generic, hipStream_t stream) {
  return gemm_bf16_impl(A, B, C, R, bias, M, N, K, lda, ldb, ldc, ldr, a_mode, b_mode, out_fp32, accumulate, alpha, force_generic, nullptr, 0,
                        stream);
}

// Same, with a caller-owned scratch buffer: lets the 256x256 kernel cut the tiles of its last, partially filled round of
// workgroups into K-slices (fp32 partials in the workspace + a fix-up pass). 64 MiB covers every split it will choose
// (<= 256 partial tiles of 256 KiB); a smaller buffer only restricts the choice. Results differ from mla_gemm_bf16 only by
// the fp32 summation order of the split tiles; the order is fixed, so the result is deterministic.
extern "C" int mla_gemm_bf16_ws(const void* A, const void* B, void* C, const void* R, const void* bias, int M, int N,
                                int K, int lda, int ldb, int ldc, int ldr, int a_mode, int b_mode, int out_fp32,
                                int accumulate, float alpha, int force_generic, float* workspace, size_t workspace_bytes,
                                hipStream_t stream) {
  MLA_CHECK_ARG(workspace == nullptr || (((uintptr_t)workspace & 15) == 0), "mla_gemm_bf16_ws: workspace must be 16-byte aligned");
  return gemm_bf16_impl(A, B, C, R, bias, M, N, K, lda, ldb, ldc, ldr, a_mode, b_mode, out_fp32, accumulate, alpha, force_generic, workspace,
                        workspace_bytes, stream);
}

// mla_gemm_bf16_ws with fp32 output that also leaves sum(C^2) of the final values as *sq_slots partial sums in sq_out (capacity in
// floats >= mla_gemm_sq_slots(M, N, K, workspace_bytes)): the gradient-norm contribution of a weight gradient without re-reading it (training/strategies/fsdp.py:
// 308-310 clips by the global norm). Fixed partial order -> deterministic. 256x256-kernel shapes only (error otherwise).
extern "C" int mla_gemm_bf16_ws_sq(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int accumulate,
                                   float alpha, float* workspace, size_t workspace_bytes, float* sq_out, int sq_capacity, int* sq_slots,
                                   hipStream_t stream) {
  MLA_CHECK_ARG(sq_out && sq_slots, "mla_gemm_bf16_ws_sq: null sq_out / sq_slots");
  MLA_CHECK_ARG(workspace == nullptr || (((uintptr_t)workspace & 15) == 0), "mla_gemm_bf16_ws_sq: workspace must be 16-byte aligned");
  return gemm_bf16_impl(A, B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, 0, 0, 1, accumulate, alpha, 0, workspace, workspace_bytes,
                        stream, sq_out, sq_capacity, sq_slots);
}

// Batched form (no bias / residual): for z = (outer, inner) in [0, n_outer) x [0, n_inner):
//   C + o*sCo + i*sCi = alpha * Aop(A + o*sAo + i*sAi) * Bop(B + o*sBo + i*sBi)^T      (strides in elements)
// Used for multi-head attention with arbitrary head_dim in the post-training generation heads
// (models/mla/generation/models.py:68-286: nn.TransformerDecoder with 8 heads of 512).
extern "C" int mla_gemm_batched_bf16(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int a_mode,
                                     int b_mode, int out_fp32, float alpha, int n_outer, int n_inner, long long sAo, long long sAi,
                                     long long sBo, long long sBi, long long sCo, long long sCi, hipStream_t stream) {
  MLA_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && n_outer > 0 && n_inner > 0, "mla_gemm_batched_bf16: bad args");
  MLA_CHECK_ARG((long long)n_outer * n_inner <= 65535, "mla_gemm_batched_bf16: too many batches");
  GemmArgs p{(const bf16_t*)A, (const bf16_t*)B, C, nullptr, nullptr, M, N, K, lda, ldb, ldc, 0, out_fp32, 0, alpha, 0,
             n_inner, sAo, sAi, sBo, sBi, sCo, sCi, 0, 0, nullptr};
  const int nb = n_outer * n_inner;
  const bool al = ((sAo | sAi | sBo | sBi) % 8 == 0) && ((sCo | sCi) % 4 == 0);
  bool mfma_ok = al && (K % 32 == 0) && (lda % 8 == 0) && (ldb % 8 == 0) && (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0) &&
                 (((uintptr_t)C & 15) == 0);
  if (a_mode == 1 && (M % 8 != 0)) mfma_ok = false;
  if (b_mode == 1 && (N % 8 != 0)) mfma_ok = false;
  if (mfma_ok) {
    if (a_mode == 0 && b_mode == 0) return launch128<0, 0>(p, stream, nb);
    if (a_mode == 0 && b_mode == 1) return launch128<0, 1>(p, stream, nb);
    if (a_mode == 1 && b_mode == 0) return launch128<1, 0>(p, stream, nb);
    return launch128<1, 1>(p, stream, nb);
  }
  dim3 grid((N + 63) / 64, (M + 63) / 64, nb);
  hipLaunchKernelGGL(gemm_generic_kernel, grid, dim3(256), 0, stream, p, a_mode, b_mode);
  MLA_LAUNCH_CHECK();
}
