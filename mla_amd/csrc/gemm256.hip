// 256x256x64 bf16 MFMA GEMM for the large Llama GEMMs (M, N >= 256, K % 64 == 0); same contract as gemm128_kernel.
//
// Why a second kernel: a 128x128 tile moves 32 KiB of operands per 2.1 MFLOP (64 flop/B) -- at the 2.5 PFLOP/s MFMA
// peak that is ~39 TB/s of L2->LDS traffic, above what the 8 XCD L2s deliver (~34.5 TB/s, MI355X_MICROARCH.md).
// 256x256 halves the operand traffic per flop (128 flop/B) and the LDS->register traffic per MFMA.
//
// Schedule (DESIGN.md "GEMM 256"):  8 waves = 2 (M) x 4 (N), per-wave output 128x64 = 8x4 MFMA fragments (128 acc VGPRs).
// A K-tile (BK = 64) lives in one of two 64 KiB LDS buffers as four 16 KiB "half-tile" slots arranged by C-quadrant:
//   SA0/SA1 = the A rows every wave needs for its upper/lower 64 output rows, SB0/SB1 = the B rows for its left/right 32
//   output columns.  Per K-tile each wave runs 4 phases = 4 quadrants of 16 MFMAs x 2 k-steps:
//     ph1: read SA0,SB0 -> Q00   ph2: read SB1 -> Q01   ph3: read SA1 -> Q11   ph4: (no read) -> Q10
//   Every phase also issues ONE half-tile of global_load_lds (2 x 16 B per lane) for a future K-tile, into the slot whose
//   last reader finished exactly 2 phases earlier (the earliest legal moment with the one-barrier stagger of the wave rows):
//   ph1: SA1(t+1)  ph2: SB0(t+2)  ph3: SA0(t+2)  ph4: SB1(t+2).  Every half-tile is then issued 6 phases (1.5 K-tiles) before
//   the phase that reads it, and the loop retires loads with a COUNTED s_waitcnt vmcnt(10): 5 half-tiles stay in flight across
//   the barriers (never vmcnt(0) in the loop). [Until r1 the order was SB1(t+1) / SA1(t+1) / SA0(t+2) / SB0(t+2) with issue-to-read
//   distances 5 / 5 / 6 / 4 phases and vmcnt(6): same LDS, 3 half-tiles in flight.] The next tile's B0 fragments are prefetched in ph4, so the
//   per-phase ds_read counts are 8/4/8/4 (ablation: LDS reads + LDS-DMA writes cost ~35 % of the MFMA-only rate).
// The two wave rows run staggered by one barrier (wave row 1 executes one extra s_barrier up front), so on every SIMD
// one wave is in its ds_read/issue segment while its partner is in its MFMA segment; s_setprio(1) wraps the MFMAs.
#include "common.h"

#include "gemm_args.h"

#include "gemm256_kloop_clobbers.inc"
// bf16 epilogue outputs (and the residual / gate|up rows the epilogues read) are touched once per launch and consumed by a LATER kernel:
// non-temporal accesses keep them from displacing the operand panels 4-8 CUs of an XCD share through its L2. Round 3, same-box A/B of
// whole steps: 587.4 -> 581.6 ms (-1.0 %), fused-epilogue kernels +4.7 % (GEMM flops over their duration); non-temporal fp32 stores
// (wgrad outputs, read by AdamW) measured 0.2 % worse and stay plain. MLA_GEMM_PLAIN_STORES restores the old behaviour (A/B builds).
#ifndef MLA_GEMM_PLAIN_STORES
#define MLA_ST16(ADDR, VAL) __builtin_nontemporal_store((u32x4_t)(VAL), (u32x4_t*)(ADDR))
#define MLA_LD16(ADDR) __builtin_nontemporal_load((const u32x4_t*)(ADDR))
#else
#define MLA_ST16(ADDR, VAL) (*(u32x4_t*)(ADDR) = (VAL))
#define MLA_LD16(ADDR) (*(const u32x4_t*)(ADDR))
#endif

namespace {

constexpr int SLOT = 16384;
constexpr int SA0 = 0, SB0 = SLOT, SB1 = 2 * SLOT, SA1 = 3 * SLOT, BUF = 4 * SLOT;

__device__ __forceinline__ int hsw(int kr) { return (kr & 3) | (((kr >> 3) & 1) << 2); }
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// global source of the 16-B chunk this lane stages for (operand, half h, wave instruction `instr` in 0..15), K-tile 0.
// Half-tile h of an operand = the CONTIGUOUS global rows/cols [base + h*128, base + h*128 + 128): quadrant QI of every
// wave lives in half QI (wave wr owns rows QI*128 + wr*64 + ..), so each LDS row of a reduction-major tile is one
// contiguous 256-B global run (split 64/128-B runs measured 1.5-2x slower on the L2->LDS path).
template <int MODE, bool IS_A>
__device__ __forceinline__ const bf16_t* stage_src(const bf16_t* g, int ld, int base, int lim, int h, int instr, int lane) {
  const int p = instr * 64 + lane;
  if (MODE == 0) {
    const int rp = p >> 3, cp = p & 7;
    const int c = cp ^ (rp & 7);
    int gi = base + h * 128 + rp;
    gi = gi < lim ? gi : lim - 1;
    return g + (size_t)gi * ld + c * 8;
  } else {
    const int kr = p >> 4, cp = p & 15;
    const int c = cp ^ (hsw(kr) << 1);
    const int ip = c * 8;
    int gi = base + h * 128 + ip;
    gi = (gi + 8 <= lim) ? gi : 0;
    return g + (size_t)kr * ld + gi;
  }
}

// EPI: 0 = plain epilogue (bias / residual / fp32 / fused RoPE by arguments), 1 = fused gate|up + SwiGLU forward, 2 = fused d(act) +
// SwiGLU backward. Separate instantiations: the fused forms are different kernels (GEMM + an HBM-bound elementwise pass in the
// epilogue) and show up under their own names in rocprofv3, so the plain kernel's statistics are not mixed with theirs.
#ifndef MLA_GROUP_M
#define MLA_GROUP_M 4   // tile rows per group of the M-grouped walk inside an XCD's range (round-2 sweep on the twelve 7B shapes: 2 is within +-1 %, 8 loses 0-2 %, 16 loses 1-5 %)
#endif
#ifndef MLA_GEMM256_UNTRACKED
#define MLA_GEMM256_UNTRACKED 0   // 1: k-contiguous instantiations stage untracked too (A/B)
#endif
template <int AMODE, int BMODE, int EPI = 0, bool ASM = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p) {
  const bool LGKM_BEFORE = false;
  const int GROUP_M = MLA_GROUP_M;
#ifdef MLA_GEMM256_ABLATION   // timing experiments only (tools/exp_dbg.py); the runtime flags cost branches in the hot loop
  const bool NO_READ = (p.debug & 4) != 0, NO_STAGE = (p.debug & 8) != 0, NO_BAR2 = (p.debug & 16) != 0, NO_BAR1 = (p.debug & 32) != 0;
  const int dbg = p.debug;
#else
  constexpr bool NO_READ = false, NO_STAGE = false, NO_BAR2 = false, NO_BAR1 = false;
  constexpr int dbg = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int li = lane & 15, lg = lane >> 4;

  const int num_m = (p.M + 255) / 256, num_n = (p.N + 255) / 256;
  // Split-K tail: workgroups [0, sk_full) compute whole tiles (XCD-remapped among themselves); workgroup sk_full + u computes
  // K-slice u % sk_split of tile sk_full + u / sk_split and leaves a raw fp32 partial tile in the workspace.
  int pid, kt0 = 0, nt = p.K / 64;
  float* part = nullptr;
  if (p.sk_split <= 1) {
    pid = xcd_remap(blockIdx.x, gridDim.x);
  } else if ((int)blockIdx.x < p.sk_full) {
    pid = xcd_remap(blockIdx.x, p.sk_full);
  } else {
    const int u = blockIdx.x - p.sk_full;
    const int slice = u % p.sk_split;
    pid = p.sk_full + u / p.sk_split;
    if ((nt & 1) == 0) {   // the assembly loop walks pairs of K-tiles: slice boundaries on even tiles (for both loops: same bits)
      const int np = nt >> 1;
      kt0 = 2 * (int)((long long)slice * np / p.sk_split);
      nt = 2 * (int)((long long)(slice + 1) * np / p.sk_split) - kt0;
    } else {
    kt0 = (int)((long long)slice * nt / p.sk_split);
    nt = (int)((long long)(slice + 1) * nt / p.sk_split) - kt0;
    }
    part = p.sk_ws + (size_t)u * 65536;
  }
#ifdef MLA_EXPERIMENTAL_KERNELS
  // Round-3 experiment (tools/exp_stagger.sh), NULL RESULT: 0 ... -2.8 % for delays of 20-70 us in either mode. The fused epilogues
  // are bound by what ONE CU can move (768 KB per tile at ~12 B/clk, the same rate with 16 workgroups on the chip as with 256), not
  // by 256 CUs bursting at once, so shifting bursts in time buys nothing; only stores issued under the SAME CU's next main loop
  // could, and 160 KB of LDS (128 KB operand ring) cannot carry a staged 128-256 KB tile alongside.
  if (EPI != 0 && p.stagger_mode && (int)blockIdx.x < 256) {     // see GemmArgs::stagger_mode
    const bool late = p.stagger_mode == 1 ? (blockIdx.x & 1) : ((blockIdx.x >> 3) & 1);
    if (late) {
      const unsigned long long t0 = wall_clock64();
      while ((long long)(wall_clock64() - t0) < (long long)p.stagger_ticks) __builtin_amdgcn_s_sleep(64);
    }
  }
#endif
  const int in_group = GROUP_M * num_n;
  const int group_id = pid / in_group;
  const int first_m = group_id * GROUP_M;
  const int gsz = (num_m - first_m) < GROUP_M ? (num_m - first_m) : GROUP_M;
  const int pid_m = first_m + (pid % in_group) % gsz;
  const int pid_n = (pid % in_group) / gsz;
  const int m0 = pid_m * 256, n0 = pid_n * 256;

  f32x4_t acc[8][4];
  if constexpr (ASM) {
    // ---- hand-scheduled main loop (gemm256_kloop.inc, generated by tools/gen_gemm_asm.py; k-contiguous operands, an even number of
    // K-tiles). Same LDS operand image, same K order and the same accumulation order per output as the compiler-scheduled loop below:
    // bit-identical results. Three barriers per K-tile instead of eight, no vmcnt(0) in the loop, and the 8 loads of a wave spread over
    // 43 of its 64 MFMA gaps (one per 6): the waves of a workgroup run in lockstep, so a burst of loads in one wave is a burst of eight
    // times as many in the CU's single address path and stalls every issuer (round 3: MfmaUtil 65 % -> 84 % on the 4-wave form of this
    // loop). The last pair of K-tiles runs in a peeled copy of the loop body that prefetches nothing.
    // The accumulators leave the assembly through LDS: the loop's tail writes the tile image the epilogues below read.
    static_assert(AMODE == 0 && BMODE == 0, "the assembly loop takes k-contiguous operands");
    const unsigned c0 = ((unsigned)lg ^ ((unsigned)li & 7u)) << 4;
    const unsigned vA0 = (unsigned)((wr * 64 + li) * 128) + c0, vB0 = 32768u + (unsigned)((wc * 32 + li) * 128) + c0;
    // staging: instruction q (0..3) of this wave writes LDS rows wave*32 + q*8 + (lane >> 3) of A and of B, lane's 16-B chunk pre-swizzled
    const unsigned lchunk = ((unsigned)lane & 7u) ^ (((unsigned)lane >> 3) & 7u);
    const int rt0 = wave * 32 + (lane >> 3);
    const int lastA = (p.M - 1 - m0) < 255 ? (p.M - 1 - m0) : 255;
    int lastB = (p.N - 1 - n0) < 255 ? (p.N - 1 - n0) : 255, bshift = 0;
    const bf16_t* pB = p.B + (size_t)n0 * p.ldb + (size_t)kt0 * 64;
    if (EPI == 1) {   // rows 0..127 of the B half-tiles = gate channels [128 pid_n, +128), rows 128..255 = the matching up channels
      pB = p.B + (size_t)pid_n * 128 * p.ldb;
      bshift = wave >= 4 ? p.sf_I - 128 : 0;
      lastB = 255;
    }
    const unsigned oA0 = (unsigned)rt0 * (unsigned)p.lda * 2u + lchunk * 16u;
    const unsigned oB0 = (unsigned)(rt0 + bshift) * (unsigned)p.ldb * 2u + lchunk * 16u;
    const unsigned oAmax = (unsigned)lastA * (unsigned)p.lda * 2u + lchunk * 16u;
    const unsigned oBmax = (unsigned)(lastB + bshift) * (unsigned)p.ldb * 2u + lchunk * 16u;
    const int sA8 = p.lda * 16, sB8 = p.ldb * 16;
    const bf16_t* pA = p.A + (size_t)m0 * p.lda + (size_t)kt0 * 64;
    const int kmax = nt * 128 - 128, nit = nt >> 1, ldsw = wave * 4096;   // kmax, nit: VGPR operands (selects of uniform values)
    const bool img_bf16 = !part && !p.out_fp32 && p.R == nullptr && p.bias == nullptr;     // uniform; mirrors the epilogue's choice below
    // RMSNorm folded into this projection (gemm_args.h rs_*): rstd of the tile's 256 rows goes to the 1 KiB behind the operand ring
    // (the loop does not touch it; the tail starts with a barrier) and the tail multiplies accumulator rows by it: emode 2
    const bool rowscale = img_bf16 && p.rs_rstd != nullptr;
    if (rowscale && tid < 256) {
      const int mr = (m0 + tid) < p.M ? (m0 + tid) : (p.M - 1);
      float r;
      if (p.rs_ss) {
        const float* sp = p.rs_ss + (size_t)mr * p.rs_parts;
        float t = 0.f;
        for (int j = 0; j < p.rs_parts; ++j) t += sp[j];
        r = 1.0f / sqrtf(t / (float)p.K + p.rs_eps);            // same form as rmsnorm_fwd_kernel (elementwise.hip)
        if (pid_n == 0 && m0 + tid < p.M) p.rs_rstd[mr] = r;
      } else {
        r = p.rs_rstd[mr];
      }
      *(float*)(smem + 2 * BUF + tid * 4) = r;
    }
    const int emode = img_bf16 ? (rowscale ? 2 : 0) : 1;       // (a VGPR operand: hipcc hands a select of uniform values to an "s" constraint in a VGPR)
    const float alpha = part ? 1.f : p.alpha;
    const unsigned vImg = img_bf16 ? (unsigned)((wr * 64 + li) * 512) + ((((unsigned)(wc * 4 + (lg >> 1))) ^ (unsigned)li) << 4) + (unsigned)(lg & 1) * 8u
                                   : (unsigned)((wr * 64 + li) * 1024) + ((((unsigned)(wc * 8 + lg)) ^ (unsigned)li) << 4);
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass would validate the register names against x86
    asm volatile(
#include "gemm256_kloop.inc"
        : : [pA] "s"(pA), [pB] "s"(pB), [kmax] "v"(kmax), [nit] "v"(nit), [ldsw] "s"(ldsw), [sA8] "s"(sA8), [sB8] "s"(sB8),
            [emode] "v"(emode), [alpha] "v"(alpha), [vA0] "v"(vA0), [vB0] "v"(vB0), [oA0] "v"(oA0), [oB0] "v"(oB0), [oAmax] "v"(oAmax),
            [oBmax] "v"(oBmax), [vImg] "v"(vImg)
        : "memory", "scc", "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", G256K_CLOBBERS);
#endif
    if constexpr (EPI == 0) if (part) {     // K-slice of a tail tile: the raw fp32 image leaves as whole rows of the tile-local [256][256] partial
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        if (hf) {
          __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
          asm volatile(
#include "gemm256_kloop_half1.inc"
              : : [alpha] "v"(alpha), [vImg] "v"(vImg) : "memory", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11",
                "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
#endif
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int row = wave * 16 + it;
          *(f32x4_t*)(part + (hf * 128 + row) * 256 + lane * 4) = *(const f32x4_t*)(smem + row * 1024 + ((lane ^ (row & 63)) << 4));
        }
      }
      return;
    }
  } else {
  // ---- staging pointers: [slot][it]; each advances one K-tile per use
  // dbg & 64: every K-tile re-loads K-tile 0 (always an L2 hit, no fabric traffic); timing / power experiments only
  const size_t stepA = (dbg & 64) ? 0 : (dbg & 2) ? 64 : (AMODE == 0 ? 64 : (size_t)64 * p.lda);
  const size_t stepB = (dbg & 64) ? 0 : (dbg & 2) ? 64 : (BMODE == 0 ? 64 : (size_t)64 * p.ldb);
  const bf16_t* pA0[2]; const bf16_t* pA1[2]; const bf16_t* pB0[2]; const bf16_t* pB1[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    if (dbg & 2) {   // EXPERIMENT: k-contiguous address pattern regardless of the mode (wrong data, timing only)
      pA0[it] = stage_src<0, true>(p.A, 4096, 0, 4096, 0, wave * 2 + it, lane);
      pA1[it] = stage_src<0, true>(p.A, 4096, 0, 4096, 1, wave * 2 + it, lane);
      pB0[it] = stage_src<0, false>(p.B, 4096, 0, 4096, 0, wave * 2 + it, lane);
      pB1[it] = stage_src<0, false>(p.B, 4096, 0, 4096, 1, wave * 2 + it, lane);
      continue;
    }
    pA0[it] = stage_src<AMODE, true>(p.A, p.lda, m0, p.M, 0, wave * 2 + it, lane);
    pA1[it] = stage_src<AMODE, true>(p.A, p.lda, m0, p.M, 1, wave * 2 + it, lane);
    if (EPI == 1) {   // fused gate|up + SwiGLU: left half-tile = 128 gate rows, right half-tile = the 128 matching up rows of the packed weight
      pB0[it] = stage_src<BMODE, false>(p.B, p.ldb, pid_n * 128, p.sf_I, 0, wave * 2 + it, lane);
      pB1[it] = stage_src<BMODE, false>(p.B, p.ldb, p.sf_I + pid_n * 128 - 128, 2 * p.sf_I, 1, wave * 2 + it, lane);
      continue;
    }
    pB0[it] = stage_src<BMODE, false>(p.B, p.ldb, n0, p.N, 0, wave * 2 + it, lane);
    pB1[it] = stage_src<BMODE, false>(p.B, p.ldb, n0, p.N, 1, wave * 2 + it, lane);
  }
  if (kt0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      pA0[it] += (size_t)kt0 * stepA; pA1[it] += (size_t)kt0 * stepA;
      pB0[it] += (size_t)kt0 * stepB; pB1[it] += (size_t)kt0 * stepB;
    }
  }
  int tA0 = 0, tA1 = 0, tB0 = 0, tB1 = 0;  // next K-tile index (relative to kt0) each slot stages (wave-uniform)
  // Reduction-major operands are read with ds_read_b64_tr_b16, which carries no memory operand: with compiler-tracked LDS-DMA copies
  // the compiler puts `s_waitcnt vmcnt(0)` in front of those reads and the ring never overlaps (this, not instruction issue, is what
  // made these instantiations slow in round 1). They stage untracked; the counted waits below order the ring as for mode 0.
  constexpr bool UNTRACKED = (AMODE != 0 || BMODE != 0) || MLA_GEMM256_UNTRACKED;

#define STAGE(PTR, TCNT, STEP, SLOTOFF)                                                           \
  do {                                                                                            \
    const size_t back__ = (TCNT < nt) ? 0 : (size_t)TCNT * (STEP); /* dummy re-load of K-tile 0 past the end */ \
    char* dst__ = smem + (TCNT & 1) * BUF + (SLOTOFF) + wave * 2048;                              \
    if (!NO_STAGE || TCNT < 2) { /* ablation: the first two K-tiles are real, so LDS holds real operands */ \
      if (UNTRACKED) {                                                                            \
        glds16_untracked(PTR[0] - back__, dst__);                                                 \
        glds16_untracked(PTR[1] - back__, dst__ + 1024);                                          \
      } else {                                                                                    \
        glds16(PTR[0] - back__, dst__);                                                           \
        glds16(PTR[1] - back__, dst__ + 1024);                                                    \
      }                                                                                           \
    }                                                                                             \
    PTR[0] += (STEP); PTR[1] += (STEP); ++TCNT;                                                   \
  } while (0)

  // ---- fragment read offsets (bytes inside a slot)
  int offA[4], offB[2];  // mode 0: [0] = ks 0 base (rb/cb added as immediates), mode 1: one per rb / cb
  if (AMODE == 0) {
    offA[0] = (wr * 64 + li) * 128 + ((lg ^ (li & 7)) * 16);
    offA[1] = offA[2] = offA[3] = 0;
  } else {
    const int h = ((li >> 2) & 3) | ((lg & 1) << 2);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
      offA[rb] = ((lg * 8 + (li >> 2)) * 16 + ((wr * 8 + rb * 2 + ((li & 3) >> 1)) ^ (h << 1))) * 16 + (li & 1) * 8;
  }
  if (BMODE == 0) {
    offB[0] = (wc * 32 + li) * 128 + ((lg ^ (li & 7)) * 16);
    offB[1] = 0;
  } else {
    const int h = ((li >> 2) & 3) | ((lg & 1) << 2);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
      offB[cb] = ((lg * 8 + (li >> 2)) * 16 + ((wc * 4 + cb * 2 + ((li & 3) >> 1)) ^ (h << 1))) * 16 + (li & 1) * 8;
  }

  auto ldA = [&](const char* slot, int rb, int ks) -> bf16x8_t {
    if (AMODE == 0) {
      return *(const bf16x8_t*)(slot + ((offA[0] + rb * 2048) ^ (ks * 64)));
    } else {
      if (dbg & 1) return *(const bf16x8_t*)(slot + (((wr * 64 + li) * 128 + ((lg ^ (li & 7)) * 16) + rb * 2048) ^ (ks * 64)));
      union { bf16x8_t v; short4_t h2[2]; } u;
      u.h2[0] = lds_tr16_b64(slot + offA[rb] + ks * 8192);
      u.h2[1] = lds_tr16_b64(slot + offA[rb] + ks * 8192 + 1024);
      return u.v;
    }
  };
  auto ldB = [&](const char* slot, int cb, int ks) -> bf16x8_t {
    if (BMODE == 0) {
      return *(const bf16x8_t*)(slot + ((offB[0] + cb * 2048) ^ (ks * 64)));
    } else {
      if (dbg & 1) return *(const bf16x8_t*)(slot + (((wc * 32 + li) * 128 + ((lg ^ (li & 7)) * 16) + cb * 2048) ^ (ks * 64)));
      union { bf16x8_t v; short4_t h2[2]; } u;
      u.h2[0] = lds_tr16_b64(slot + offB[cb] + ks * 8192);
      u.h2[1] = lds_tr16_b64(slot + offB[cb] + ks * 8192 + 1024);
      return u.v;
    }
  };

#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];
  bool rd_on = true;   // ablation NO_READ: fragments are read for the first two K-tiles only (registers keep real operands)

#define READ_A(SLOTP)                                              \
  if (rd_on) _Pragma("unroll") for (int rb = 0; rb < 4; ++rb) {    \
    fa[rb][0] = ldA(SLOTP, rb, 0); fa[rb][1] = ldA(SLOTP, rb, 1);  \
  }
#define READ_B(DST, SLOTP)                                          \
  if (rd_on) _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) {     \
    DST[cb][0] = ldB(SLOTP, cb, 0); DST[cb][1] = ldB(SLOTP, cb, 1); \
  }
#define SEG_END()                                                  \
  __builtin_amdgcn_sched_barrier(0);                               \
  if (LGKM_BEFORE) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); \
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");           \
  if (!NO_BAR1) __builtin_amdgcn_s_barrier();                      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               \
  __builtin_amdgcn_sched_barrier(0);
#define MMA(QI, QJ, FB)                                                                                          \
  __builtin_amdgcn_s_setprio(1);                                                                                 \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
  _Pragma("unroll") for (int rb = 0; rb < 4; ++rb)                                                               \
  _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)                                                               \
    acc[QI * 4 + rb][QJ * 2 + cb] =                                                                              \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[cb][ks], fa[rb][ks], acc[QI * 4 + rb][QJ * 2 + cb], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                                             \
  asm volatile("" ::: "memory");                                                                                 \
  if (!NO_BAR2) __builtin_amdgcn_s_barrier();                                                                    \
  asm volatile("" ::: "memory");

  // ---- prologue: K-tiles 0 (all four slots) and 1 (SA0, SB0, SB1; SA1(1) is staged in phase 1 of tile 0)
  // issue order = the loop's order (SB0, SA0, SB1, SA1 per K-tile): vmcnt is an in-order counter, the counted waits rely on it
  STAGE(pB0, tB0, stepB, SB0);
  STAGE(pA0, tA0, stepA, SA0);
  STAGE(pB1, tB1, stepB, SB1);
  STAGE(pA1, tA1, stepA, SA1);
  STAGE(pB0, tB0, stepB, SB0);
  STAGE(pA0, tA0, stepA, SA0);
  STAGE(pB1, tB1, stepB, SB1);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // SB0(0), SA0(0) of this wave have landed
  __builtin_amdgcn_s_barrier();
  READ_B(fb0, smem + SB0)                            // B0 fragments of K-tile 0 (later tiles prefetch theirs in phase 4)
  if (wr == 1) __builtin_amdgcn_s_barrier();         // stagger wave row 1 by one barrier

  // One K-tile. X holds the B0 fragments of this tile on entry; Y receives B1 (phase 2) and then the NEXT tile's B0
  // (phase 4), so the fragment reads are spread 8 / 4 / 8 / 4 over the four phases instead of 12 / 4 / 8 / 0.
#define TILE_BODY(X, Y, T_)                                          \
  {                                                                  \
    const char* buf = smem + ((T_) & 1) * BUF;                       \
    const char* nbuf = smem + (((T_) + 1) & 1) * BUF;                \
    if (NO_READ) rd_on = (T_) < 2;                                   \
    STAGE(pA1, tA1, stepA, SA1);   /* SA1(t+1) */                    \
    READ_A(buf + SA0)                                                \
    SEG_END()                                                        \
    MMA(0, 0, X)                                                     \
    STAGE(pB0, tB0, stepB, SB0);   /* SB0(t+2) */                    \
    READ_B(Y, buf + SB1)                                             \
    SEG_END()                                                        \
    MMA(0, 1, Y)                                                     \
    STAGE(pA0, tA0, stepA, SA0);   /* SA0(t+2) */                    \
    READ_A(buf + SA1)                                                \
    SEG_END()                                                        \
    MMA(1, 1, Y)                                                     \
    STAGE(pB1, tB1, stepB, SB1);   /* SB1(t+2) */                    \
    READ_B(Y, nbuf + SB0)                                            \
    SEG_END()                                                        \
    MMA(1, 0, X)                                                     \
  }
  int t = 0;
  for (; t + 1 < nt; t += 2) {
    TILE_BODY(fb0, fb1, t)
    TILE_BODY(fb1, fb0, t + 1)
  }
  if (t < nt) TILE_BODY(fb0, fb1, t)
#undef TILE_BODY
  if (wr == 0) __builtin_amdgcn_s_barrier();          // re-balance the stagger
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // drain the dummy tail loads before the wave retires
  }   // !ASM

  // ---- epilogue (operands were passed swapped: lane holds C[m][n..n+3])
  if constexpr (!ASM) if (part) {      // K-slice of a tail tile: raw accumulators, tile-local [256][256] fp32; the fix-up kernel finishes the job
#pragma unroll
    for (int ri = 0; ri < 8; ++ri) {
      const int ml = (ri >> 2) * 128 + wr * 64 + (ri & 3) * 16 + li;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const int nl = (ci >> 1) * 128 + wc * 32 + (ci & 1) * 16 + lg * 4;
        *(f32x4_t*)(part + ml * 256 + nl) = acc[ri][ci];
      }
    }
    return;
  }
  const bool vec_ok = ((p.ldc & 3) == 0) && (p.R == nullptr || (p.ldr & 3) == 0);
  // ---- fast epilogue: the tile goes through LDS (free now) and leaves as WHOLE ROWS -- 16 B per lane, 512 B / 1 KiB contiguous
  // per half-wave / wave. The direct path below stores 8 B (bf16) per lane to 16 different rows per instruction (32-B pieces of 16
  // cache lines): measured 8-13 us per tile round, i.e. 10-20 % of a K = 4096 launch (tools/exp_epilogue.py).
  //   bf16 output without bias / residual: one pass, bf16 image [256][256] (row pitch 512 B, 16-B chunks XOR-swizzled by row & 31)
  //   fp32 output, or bias / residual (added in fp32 before the single rounding): two passes of 128 rows, fp32 image
  //   [128][256] (row pitch 1 KiB, chunks swizzled by row & 63)
  const bool fast = (p.N & 7) == 0 && (p.ldc & 7) == 0 && (((uintptr_t)p.C) & 15) == 0 &&
                    (p.R == nullptr || ((p.ldr & 7) == 0 && (((uintptr_t)p.R) & 15) == 0)) &&
                    (p.bias == nullptr || (((uintptr_t)p.bias) & 15) == 0);
  if (fast) {
    __syncthreads();     // every wave is done with the operand image (and has drained its own LDS-DMA loads above)
    if (!p.out_fp32 && p.R == nullptr && p.bias == nullptr) {
      if constexpr (!ASM) {     // (the assembly loop's tail has written this image already)
      if (p.rs_rstd) {          // RMSNorm folded into this projection: rstd of the tile's rows, staged behind the ring (see the assembly path)
        if (tid < 256) {
          const int mr = (m0 + tid) < p.M ? (m0 + tid) : (p.M - 1);
          float r;
          if (p.rs_ss) {
            const float* sp = p.rs_ss + (size_t)mr * p.rs_parts;
            float t = 0.f;
            for (int j = 0; j < p.rs_parts; ++j) t += sp[j];
            r = 1.0f / sqrtf(t / (float)p.K + p.rs_eps);
            if (pid_n == 0 && m0 + tid < p.M) p.rs_rstd[mr] = r;
          } else {
            r = p.rs_rstd[mr];
          }
          *(float*)(smem + 2 * BUF + tid * 4) = r;
        }
        __syncthreads();
      }
#pragma unroll
      for (int ri = 0; ri < 8; ++ri) {
        const int ml = (ri >> 2) * 128 + wr * 64 + (ri & 3) * 16 + li;
        const float rsc = p.rs_rstd ? *(const float*)(smem + 2 * BUF + ml * 4) : p.alpha;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          const int nl = (ci >> 1) * 128 + wc * 32 + (ci & 1) * 16 + lg * 4;
          u32x2_t o;
          o[0] = pack2bf(acc[ri][ci][0] * rsc, acc[ri][ci][1] * rsc);
          o[1] = pack2bf(acc[ri][ci][2] * rsc, acc[ri][ci][3] * rsc);
          *(u32x2_t*)(smem + ml * 512 + ((((nl >> 3) ^ (ml & 31))) << 4) + ((nl >> 2) & 1) * 8) = o;
        }
      }
      }
      __syncthreads();
      if (EPI == 1) {
        // ---- fused SwiGLU forward (mla_gemm_gateup_swiglu): image columns 0..127 = gate, 128..255 = up of channels [128 pid_n, +128).
        // Per strip of 64 tokens: gate|up rows leave as two 256-B runs per row; act = swiglu_fwd_elem(g, u) is stored row-major and --
        // when the caller keeps it for the backward -- goes back into the strip's own (consumed) image rows as [channel][token] lines
        // (token slot XORed with 2 x piece index) and leaves as 16-B pieces of 8 tokens, like the backward epilogue below.
        const int I = p.sf_I, cbase = pid_n * 128;
        const size_t ld2 = (size_t)2 * I;
        for (int s4 = 0; s4 < 4; ++s4) {
          char* scrA = smem + s4 * 64 * 512;
          u32x4_t pa[2];
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const int idx = tid + 512 * k4;
            const int rl = idx >> 5, ch = idx & 31;
            const int row = s4 * 64 + rl, m = m0 + row;
            if (p.C && m < p.M)      // (null: the forward of a checkpointed layer keeps nothing but its input; gate|up is recomputed later)
              MLA_ST16((bf16_t*)p.C + (size_t)m * ld2 + (ch < 16 ? cbase + ch * 8 : I + cbase + (ch - 16) * 8), *(const u32x4_t*)(smem + row * 512 + ((ch ^ (row & 31)) << 4)));
          }
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            const int idx = tid + 512 * k2;
            const int rl = idx >> 4, c = idx & 15;
            const int row = s4 * 64 + rl, m = m0 + row;
            float gv[8], uv[8], o[8];
            unpack8(*(const u32x4_t*)(smem + row * 512 + ((c ^ (row & 31)) << 4)), gv);
            unpack8(*(const u32x4_t*)(smem + row * 512 + (((16 + c) ^ (row & 31)) << 4)), uv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = swiglu_fwd_elem(gv[j], uv[j]);
            pa[k2] = pack8(o);
            if (p.sf_act && m < p.M) MLA_ST16(p.sf_act + (size_t)m * I + cbase + c * 8, pa[k2]);     // (null: the caller wants act^T only)
          }
          if (p.sf_actT == nullptr) continue;          // uniform: the caller does not keep the transposed product
          __syncthreads();
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            const int idx = tid + 512 * k2;
            const int rl = idx >> 4, c = idx & 15;
            const int slot2 = (rl ^ (2 * c)) * 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int c0 = (c * 8 + 2 * j) * 128 + slot2;
              *(unsigned short*)(scrA + c0) = (unsigned short)(pa[k2][j] & 0xffffu);
              *(unsigned short*)(scrA + c0 + 128) = (unsigned short)(pa[k2][j] >> 16);
            }
          }
          __syncthreads();
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) {
            const int idx = tid + 512 * k2;
            const int col = idx >> 3, rc = idx & 7;          // 128 channel lines x 8 pieces of 8 tokens
            const int chc = col >> 3, b = chc & 3;
            const int m = m0 + s4 * 64 + rc * 8;
            if (m < p.M) {
              u32x4_t w = *(const u32x4_t*)(scrA + col * 128 + ((rc ^ (chc >> 2)) << 4));
              if (b & 1) w = u32x4_t{w[1], w[0], w[3], w[2]};
              if (b & 2) w = u32x4_t{w[2], w[3], w[0], w[1]};
              MLA_ST16(p.sf_actT + (size_t)(cbase + col) * p.sf_ldt + m, w);
            }
          }
        }
        return;
      }
      if (EPI == 2) {
        // ---- fused SwiGLU backward (mla_gemm_dact_swiglu_bwd): the staged tile is d(act)[256 tokens][256 channels], rounded to bf16
        // exactly like the stand-alone GEMM would have stored it. Four strips of 64 tokens: (a) every thread takes four 8-channel
        // pieces, reads gate / up from HBM (whole 512-B row runs), applies swiglu_bwd_elem, stores d(gate) / d(up) row-major;
        // (b) the same values go back into LDS TRANSPOSED -- into the strip's own, now consumed rows of the image and into 32 KiB
        // beyond the operand ring -- as [channel][token] lines of 128 B whose token slot is XORed with 2 x piece index (bank-
        // conflict-free 2-byte writes); (c) the lines are read back as 16-B pieces of 8 tokens (undoing the XOR: piece index and a
        // dword permutation) and stored to d(gate|up)^T. d(act) costs no HBM traffic at all (was: 1 write + 1 read per element).
        // Where its time goes (round 3, ablation build, per launch at the 7B shape; 1 555 us with the full epilogue, 1 094 us with
        // none, the plain GEMM of the same shape incl. its own epilogue 1 153 us): gate|up loads 88 us, row-major stores 107, transposed
        // stores 116, LDS transposition 27-40, arithmetic (one full-precision divide per element) + image reads + barriers ~120 --
        // additive. Each 256-KB-per-tile stream costs what 772 MB cost at ~7.5 TB/s: all 256 CUs run their epilogues at the same
        // time and share the memory system, so the I/O is already at the chip's rate and only overlap with OTHER workgroups' main
        // loops could hide it. Tried and measured: gate|up fetched one strip ahead (-0.5 %, noise), all 32 loads up front from
        // inline assembly with counted waits (+2 %: 256 KB of reads per CU queue in front of the stores), first-round stagger (0 ... -3 %).
        const int I = p.sw_I;
        char* scrB = smem + 2 * BUF;
        const size_t ld2 = (size_t)2 * I;
        // Assembly-loop kernels fetch gate / up ONE STRIP AHEAD (the loop below stays a loop, so the values are carried in registers and
        // the compiler cannot sink the loads to their use as it did in the compiler-loop kernel, where 128 accumulator registers are
        // live): the load latency of strips 1..3 disappears behind the previous strip's arithmetic, stores and LDS transposition.
        u32x4_t gq[4], uq[4];
        auto fetch_gu = [&](int strip, u32x4_t* g4, u32x4_t* u4) {
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const int idx = tid + 512 * k4;
            const int m = m0 + strip * 64 + (idx >> 5), n = n0 + (idx & 31) * 8;
            g4[k4] = u32x4_t{0u, 0u, 0u, 0u};
            u4[k4] = g4[k4];
            if (m < p.M && n < p.N) {
              g4[k4] = MLA_LD16(p.sw_gu + (size_t)m * ld2 + n);
              u4[k4] = MLA_LD16(p.sw_gu + (size_t)m * ld2 + I + n);
            }
          }
        };
        if constexpr (ASM) fetch_gu(0, gq, uq);
        for (int s4 = 0; s4 < 4; ++s4) {
          char* scrA = smem + s4 * 64 * 512;
          u32x4_t pg[4], pu[4];
          u32x4_t gn[4], un[4];
          if constexpr (ASM) if (s4 < 3) fetch_gu(s4 + 1, gn, un);
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const int idx = tid + 512 * k4;
            const int rl = idx >> 5, ch = idx & 31;
            const int row = s4 * 64 + rl;
            const int m = m0 + row, n = n0 + ch * 8;
            pg[k4] = u32x4_t{0u, 0u, 0u, 0u};
            pu[k4] = pg[k4];
            if (m < p.M && n < p.N) {
              float dv[8], gv[8], uv[8], dg[8], du[8];
              unpack8(*(const u32x4_t*)(smem + row * 512 + ((ch ^ (row & 31)) << 4)), dv);
              if constexpr (ASM) {
                unpack8(gq[k4], gv);
                unpack8(uq[k4], uv);
              } else {
              unpack8(*(const u32x4_t*)(p.sw_gu + (size_t)m * ld2 + n), gv);
              unpack8(*(const u32x4_t*)(p.sw_gu + (size_t)m * ld2 + I + n), uv);
              }
#pragma unroll
              for (int j = 0; j < 8; ++j) swiglu_bwd_elem(dv[j], gv[j], uv[j], dg[j], du[j]);
              pg[k4] = pack8(dg);
              pu[k4] = pack8(du);
              MLA_ST16(p.sw_dgu + (size_t)m * ld2 + n, pg[k4]);
              MLA_ST16(p.sw_dgu + (size_t)m * ld2 + I + n, pu[k4]);
            }
          }
          __syncthreads();           // the strip's rows of the d(act) image have been consumed by everyone
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const int idx = tid + 512 * k4;
            const int rl = idx >> 5, ch = idx & 31;
            const int slot2 = (rl ^ (2 * ch)) * 2;                      // byte offset of the token slot inside a 128-B channel line
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int c0 = (ch * 8 + 2 * j) * 128 + slot2;
              *(unsigned short*)(scrA + c0) = (unsigned short)(pg[k4][j] & 0xffffu);
              *(unsigned short*)(scrA + c0 + 128) = (unsigned short)(pg[k4][j] >> 16);
              *(unsigned short*)(scrB + c0) = (unsigned short)(pu[k4][j] & 0xffffu);
              *(unsigned short*)(scrB + c0 + 128) = (unsigned short)(pu[k4][j] >> 16);
            }
          }
          __syncthreads();
#pragma unroll
          for (int k4 = 0; k4 < 4; ++k4) {
            const int idx = tid + 512 * k4;
            const int col = idx >> 3, rc = idx & 7;
            const int chc = col >> 3;                                   // the piece index the writer XORed with
            const int off = col * 128 + ((rc ^ (chc >> 2)) << 4);
            const int b = chc & 3;
            const int m = m0 + s4 * 64 + rc * 8, n = n0 + col;
            if (m < p.M && n < p.N) {
#pragma unroll
              for (int which = 0; which < 2; ++which) {
                u32x4_t w = *(const u32x4_t*)((which ? scrB : scrA) + off);
                if (b & 1) w = u32x4_t{w[1], w[0], w[3], w[2]};
                if (b & 2) w = u32x4_t{w[2], w[3], w[0], w[1]};
                MLA_ST16(p.sw_dguT + (size_t)(which * I + n) * p.sw_ldt + m, w);
              }
            }
          }
          // no barrier here: the next strip's compute phase touches only ITS rows of the image and ends with a barrier before
          // anything is written to scrB / its scrA again
          if constexpr (ASM) {
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) { gq[k4] = gn[k4]; uq[k4] = un[k4]; }
          }
        }
        return;
      }
      const int ch = lane & 31;
      const int n = n0 + ch * 8;
      if (p.rope_cos && n0 < p.rope_cols) {
        // fused rotary embedding (apply_rotary_pos_emb, modeling_llama.py:184-208) on the bf16-rounded projection: the tile is two
        // whole heads; a lane's 8 columns pair with the chunk 8 positions away in the same row (d <-> d + 64). Same arithmetic as
        // rope_kernel (elementwise.hip), so the result is bit-identical to GEMM + mla_rope_inplace.
        if (n < p.N) {
          const bool lo = (ch & 8) == 0;
#pragma unroll 4
          for (int it = 0; it < 16; ++it) {
            const int row = wave * 32 + it * 2 + (lane >> 5);
            const int m = m0 + row;
            if (m < p.M) {
              float own[8], par[8], c[8], sn[8], o[8];
              unpack8(*(const u32x4_t*)(smem + row * 512 + ((ch ^ (row & 31)) << 4)), own);
              unpack8(*(const u32x4_t*)(smem + row * 512 + (((ch ^ 8) ^ (row & 31)) << 4)), par);
              const float* ct = p.rope_cos + (size_t)(m % p.rope_S) * 64 + (ch & 7) * 8;
              const float* st = p.rope_sin + (size_t)(m % p.rope_S) * 64 + (ch & 7) * 8;
              const f32x4_t c0 = *(const f32x4_t*)ct, c1 = *(const f32x4_t*)(ct + 4), s0 = *(const f32x4_t*)st, s1 = *(const f32x4_t*)(st + 4);
#pragma unroll
              for (int j = 0; j < 4; ++j) { c[j] = c0[j]; c[j + 4] = c1[j]; sn[j] = s0[j]; sn[j + 4] = s1[j]; }
#pragma unroll
              for (int j = 0; j < 8; ++j)
                o[j] = lo ? fmaf(own[j], c[j], -(par[j] * sn[j])) : fmaf(own[j], c[j], par[j] * sn[j]);
              MLA_ST16((bf16_t*)p.C + (size_t)m * p.ldc + n, pack8(o));
            }
          }
        }
        return;
      }
      if (n < p.N) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int row = wave * 32 + it * 2 + (lane >> 5);
          const int m = m0 + row;
          if (m < p.M)
            MLA_ST16((bf16_t*)p.C + (size_t)m * p.ldc + n, *(const u32x4_t*)(smem + row * 512 + ((ch ^ (row & 31)) << 4)));
        }
      }
      return;
    }
    if constexpr (EPI == 0 || !ASM) {   // the fused-epilogue assembly-loop kernels have the bf16 image only: ONE inline-asm statement, no
                                        // accumulator outlives it (tools/check_kloop_asm.py looks between the first and the last statement)
    float sq = 0.f;                                // sum of squares of this lane's FINAL fp32 outputs (p.sq_out)
    // bf16 output + residual (o / down projection of the decoder layer): the residual rows of BOTH halves are requested here, before the
    // barriers and the image reads they would otherwise queue behind (assembly-loop kernels: the accumulators are not in compiler
    // registers, so 64 registers of prefetch are free). Same values, same arithmetic.
    u32x4_t rpre[2][8];
    const bool r_pre = ASM && p.R != nullptr && !p.out_fp32;
    if (r_pre) {
      const int nn = n0 + (lane & 31) * 8;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int m = m0 + h2 * 128 + wave * 16 + it * 2 + (lane >> 5);
          rpre[h2][it] = u32x4_t{0u, 0u, 0u, 0u};
          if (m < p.M && nn < p.N) rpre[h2][it] = MLA_LD16(p.R + (size_t)m * p.ldr + nn);
        }
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (hf) __syncthreads();                     // the readers of the first half are done
      if constexpr (ASM) {    // the first half was written by the loop's tail; the second is still in the accumulator registers, which
                              // nothing between the two statements touches (build.sh checks the kernel's disassembly for that)
#if defined(__HIP_DEVICE_COMPILE__)
        if (hf) {
          const unsigned vImg2 = (unsigned)((wr * 64 + li) * 1024) + ((((unsigned)(wc * 8 + lg)) ^ (unsigned)li) << 4);
          asm volatile(
#include "gemm256_kloop_half1.inc"
              : : [alpha] "v"(p.alpha), [vImg] "v"(vImg2) : "memory", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10",
                "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
        }
#endif
      } else {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int ri = hf * 4 + r4;
        const int ml = wr * 64 + r4 * 16 + li;     // row inside this half
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          const int nl = (ci >> 1) * 128 + wc * 32 + (ci & 1) * 16 + lg * 4;
          *(f32x4_t*)(smem + ml * 1024 + (((nl >> 2) ^ (ml & 63)) << 4)) = acc[ri][ci] * p.alpha;
        }
      }
      }
      __syncthreads();
      if (p.out_fp32) {
        const int n = n0 + lane * 4;
        if (n < p.N) {
#pragma unroll
          for (int it = 0; it < 16; ++it) {
            const int row = wave * 16 + it;
            const int m = m0 + hf * 128 + row;
            if (m < p.M) {
              f32x4_t o = *(const f32x4_t*)(smem + row * 1024 + ((lane ^ (row & 63)) << 4));
              if (p.bias) {
                const u32x2_t bb = *(const u32x2_t*)(p.bias + n);
                o[0] += bflo(bb[0]); o[1] += bfhi(bb[0]); o[2] += bflo(bb[1]); o[3] += bfhi(bb[1]);
              }
              if (p.R) {
                const u32x2_t rr = *(const u32x2_t*)(p.R + (size_t)m * p.ldr + n);
                o[0] += bflo(rr[0]); o[1] += bfhi(rr[0]); o[2] += bflo(rr[1]); o[3] += bfhi(rr[1]);
              }
              float* c = (float*)p.C + (size_t)m * p.ldc + n;
              if (p.accumulate) o += *(const f32x4_t*)c;
              *(f32x4_t*)c = o;
              sq += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
            }
          }
        }
      } else {
        const int ch = lane & 31;
        const int n = n0 + ch * 8;
        if (n < p.N) {
          float bv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) bv[j] = 0.f;
          if (p.bias) unpack8(*(const u32x4_t*)(p.bias + n), bv);
          float ngv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) ngv[j] = 0.f;
          if (p.nrm_xg) unpack8(*(const u32x4_t*)(p.nrm_g + n), ngv);
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int row = wave * 16 + it * 2 + (lane >> 5);
            const int m = m0 + hf * 128 + row;
            if (m < p.M) {
              const f32x4_t a = *(const f32x4_t*)(smem + row * 1024 + (((2 * ch) ^ (row & 63)) << 4));
              const f32x4_t b = *(const f32x4_t*)(smem + row * 1024 + (((2 * ch + 1) ^ (row & 63)) << 4));
              float v[8] = {a[0] + bv[0], a[1] + bv[1], a[2] + bv[2], a[3] + bv[3], b[0] + bv[4], b[1] + bv[5], b[2] + bv[6], b[3] + bv[7]};
              if (p.R) {
                float rv[8];
                unpack8(r_pre ? rpre[hf][it] : *(const u32x4_t*)(p.R + (size_t)m * p.ldr + n), rv);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += rv[j];
              }
              const u32x4_t hq = pack8(v);
              MLA_ST16((bf16_t*)p.C + (size_t)m * p.ldc + n, hq);
              if (p.nrm_xg) {      // uniform: the NEXT RMSNorm's column scale and sum of squares leave with the rows (gemm_args.h nrm_*)
                float hv[8], xg[8];
                unpack8(hq, hv);
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { q += hv[j] * hv[j]; xg[j] = hv[j] * ngv[j]; }
                MLA_ST16(p.nrm_xg + (size_t)m * p.ldc + n, pack8(xg));
                q = half_wave_sum_last(q);       // the 32 lanes of this row (N % 256 == 0: all of them are inside the matrix)
                if (ch == 31) p.nrm_ss[(size_t)m * num_n + pid_n] = q;
              }
            }
          }
        }
      }
    }
    if (p.sq_out) {          // uniform; fixed reduction order (lanes, then the 8 waves): deterministic
      sq = wave_sum(sq);
      __syncthreads();
      if (lane == 0) ((float*)smem)[wave] = sq;
      __syncthreads();
      if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += ((float*)smem)[w];
        p.sq_out[pid] = t;
      }
    }
    }
    return;
  }
  if constexpr (!ASM) {  // (the dispatcher gives the assembly-loop instantiations fast-epilogue launches only)
#pragma unroll
  for (int ri = 0; ri < 8; ++ri) {
    const int m = m0 + (ri >> 2) * 128 + wr * 64 + (ri & 3) * 16 + li;
    if (m >= p.M) continue;

#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      const int n = n0 + (ci >> 1) * 128 + wc * 32 + (ci & 1) * 16 + lg * 4;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[ri][ci][r] * p.alpha;
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
          if (p.accumulate) o += *(const f32x4_t*)c;
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
#undef STAGE
#undef READ_A
#undef READ_B
#undef SEG_END
#undef MMA
}

#ifdef MLA_EXPERIMENTAL_KERNELS   // opt-in experiment (build.sh: MLA_EXPERIMENTAL=1), never faster than the one-tile launch
// ------------------------------------------------------------------------------------------------ persistent variant (NT)
// One workgroup per CU walks a static list of work units (whole tiles, then at most a few split-K tail slices) with the K-tile
// ring running CONTINUOUSLY across units: while the last K-tiles of unit u are multiplied, the staging cursors are already
// loading the first K-tiles of unit u+1, so the pipeline fill of every tile but the first is hidden, and there is no workgroup
// launch between tiles (measured on the non-persistent kernel: ~19 us per tile at K = 4096, 19 % of a 64-K-tile tile).
// Staging: global_load_lds with a uniform base + 32-bit lane offsets; what changes from K-tile to K-tile and from unit to unit is
// the scalar base, the lane offsets are recomputed only at a unit switch (row clamp at the M / N edge).
// Unit order: XCD x (= blockIdx & 7, the hardware's round-robin) owns a contiguous range of tile ids, its W = gridDim / 8
// workgroups take ids start + j + W * i, i.e. at any time the XCD works on W consecutive ids (a GROUP_M x W/GROUP_M block
// of tiles sharing operand panels in its L2), exactly like the remapped non-persistent launch.
struct Unit {
  int m0, n0, kt0, nt;
  float* part;
};

__global__ __launch_bounds__(512, 2) void gemm256p_kernel(GemmArgs p) {
  const int GROUP_M = MLA_GROUP_M;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int li = lane & 15, lg = lane >> 4;
  const int num_m = (p.M + 255) / 256, num_n = (p.N + 255) / 256;
  const int ntK = p.K / 64;

  // ---- this workgroup's unit list
  const int G = gridDim.x, W = G >> 3, b = blockIdx.x, x = b & 7, j = b >> 3;
  const int F = p.sk_split > 1 ? p.sk_full : num_m * num_n;           // whole tiles
  const int q8 = F >> 3, r8 = F & 7;
  const int cnt_x = q8 + (x < r8 ? 1 : 0);
  const int start_x = x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8;
  const int nfull = j < cnt_x ? (cnt_x - j + W - 1) / W : 0;
  const int tail_units = p.sk_split > 1 ? (num_m * num_n - p.sk_full) * p.sk_split : 0;
  const int ntail = b < tail_units ? (tail_units - b + G - 1) / G : 0;
  const int n_units = nfull + ntail;
  if (n_units == 0) return;
  auto get_unit = [&](int i) -> Unit {
    Unit u;
    int pid;
    if (i < nfull) {
      pid = start_x + j + W * i;
      u.kt0 = 0; u.nt = ntK; u.part = nullptr;
    } else {
      const int t = b + (i - nfull) * G;
      const int slice = t % p.sk_split;
      pid = p.sk_full + t / p.sk_split;
      u.kt0 = (int)((long long)slice * ntK / p.sk_split);
      u.nt = (int)((long long)(slice + 1) * ntK / p.sk_split) - u.kt0;
      u.part = p.sk_ws + (size_t)t * 65536;
    }
    const int in_group = GROUP_M * num_n;
    const int first_m = (pid / in_group) * GROUP_M;
    const int gsz = (num_m - first_m) < GROUP_M ? (num_m - first_m) : GROUP_M;
    u.m0 = __builtin_amdgcn_readfirstlane((first_m + (pid % in_group) % gsz) * 256);
    u.n0 = __builtin_amdgcn_readfirstlane(((pid % in_group) / gsz) * 256);
    u.kt0 = __builtin_amdgcn_readfirstlane(u.kt0);
    u.nt = __builtin_amdgcn_readfirstlane(u.nt);
    return u;
  };

  // ---- staging: global_load_lds in its (uniform base + 32-bit lane offset) form. Per slot X in {A0, B0, B1, A1}: gb = uniform
  // source address of the next K-tile of that half-tile, vo[2] = this lane's byte offsets inside the half-tile (rows clamped to the
  // last valid one at the M / N edge), left = K-tiles left in the slot's current unit, su = its unit index, t = running K-tile
  // count (LDS buffer parity). [The first version staged through buffer_load ... lds with SGPR offsets: that form costs the issuing
  // wave 100-230 cycles per load (tools/micro/dma_asm.hip) and made the K-tile slope 4-5 % worse than the one-tile kernel's.]
  const int rowA = p.lda * 2, rowB = p.ldb * 2;      // bytes per operand row
  const char* gbA0; const char* gbA1; const char* gbB0; const char* gbB1;
  unsigned voA0[2], voA1[2], voB0[2], voB1[2];
  int stA0 = 128, stA1 = 128, stB0 = 128, stB1 = 128;   // K-tile byte step; 0 once a slot has run past its last unit
  int leftA0, leftB0, leftB1, leftA1, suA0 = 0, suB0 = 0, suB1 = 0, suA1 = 0, tA0 = 0, tB0 = 0, tB1 = 0, tA1 = 0;
  // points slot state at unit u: half-tile rows [hb, hb + 128) of the operand, K-tile u.kt0
#define PSLOT_SET(X, U, ISA, HALF)                                                                                  \
  do {                                                                                                              \
    const int lim_rows__ = (ISA) ? p.M : p.N;                                                                       \
    int hb__ = ((ISA) ? (U).m0 : (U).n0) + (HALF) * 128;                                                            \
    int lim__ = lim_rows__ - 1 - hb__;                                                                              \
    if (lim__ < 0) { hb__ = lim_rows__ - 128; lim__ = 127; }      /* half-tile entirely past the edge: any valid rows */ \
    lim__ = lim__ < 127 ? lim__ : 127;                                                                              \
    gb##X = (const char*)((ISA) ? p.A : p.B) + (size_t)hb__ * ((ISA) ? rowA : rowB) + (size_t)(U).kt0 * 128;        \
    _Pragma("unroll") for (int it__ = 0; it__ < 2; ++it__) {                                                        \
      const int q__ = (wave * 2 + it__) * 64 + lane;                                                                \
      const int rp__ = q__ >> 3, cp__ = q__ & 7;                                                                    \
      const int c__ = cp__ ^ (rp__ & 7);                                                                            \
      const int re__ = rp__ < lim__ ? rp__ : lim__;                                                                 \
      vo##X[it__] = (unsigned)((re__ * ((ISA) ? p.lda : p.ldb) + c__ * 8) * 2);                                     \
    }                                                                                                               \
    left##X = (U).nt;                                                                                               \
  } while (0)
  Unit cu = get_unit(0);
  PSLOT_SET(A0, cu, true, 0);
  PSLOT_SET(A1, cu, true, 1);
  PSLOT_SET(B0, cu, false, 0);
  PSLOT_SET(B1, cu, false, 1);

#define PSTAGE(X, ISA, HALF, SLOTOFF)                                                                                      \
  do {                                                                                                                     \
    char* dst__ = smem + (t##X & 1) * BUF + (SLOTOFF) + wave * 2048;                                                       \
    glds16(gb##X + vo##X[0], dst__);                                                                                       \
    glds16(gb##X + vo##X[1], dst__ + 1024);                                                                                \
    gb##X += st##X; ++t##X;                                                                                                \
    if (--left##X == 0) {                                                                                                  \
      if (++su##X < n_units) {                                                                                             \
        const Unit nu__ = get_unit(su##X);                                                                                 \
        PSLOT_SET(X, nu__, ISA, HALF);                                                                                     \
      } else {                                                                                                             \
        /* past the last unit: keep re-loading the last K-tile of the last unit into dead slots */                        \
        gb##X -= 128; st##X = 0; left##X = 0x7fffffff;                                                                     \
      }                                                                                                                    \
    }                                                                                                                      \
  } while (0)
#define ST_A0() PSTAGE(A0, true, 0, SA0)
#define ST_A1() PSTAGE(A1, true, 1, SA1)
#define ST_B0() PSTAGE(B0, false, 0, SB0)
#define ST_B1() PSTAGE(B1, false, 1, SB1)

  // ---- fragment read offsets (bytes inside a slot)
  const int offA = (wr * 64 + li) * 128 + ((lg ^ (li & 7)) * 16);
  const int offB = (wc * 32 + li) * 128 + ((lg ^ (li & 7)) * 16);
  auto ldA = [&](const char* slot, int rb, int ks) -> bf16x8_t { return *(const bf16x8_t*)(slot + ((offA + rb * 2048) ^ (ks * 64))); };
  auto ldB = [&](const char* slot, int cb, int ks) -> bf16x8_t { return *(const bf16x8_t*)(slot + ((offB + cb * 2048) ^ (ks * 64))); };

  f32x4_t acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];

#define PREAD_A(SLOTP)                                             \
  _Pragma("unroll") for (int rb = 0; rb < 4; ++rb) {               \
    fa[rb][0] = ldA(SLOTP, rb, 0); fa[rb][1] = ldA(SLOTP, rb, 1);  \
  }
#define PREAD_B(DST, SLOTP)                                         \
  _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) {                \
    DST[cb][0] = ldB(SLOTP, cb, 0); DST[cb][1] = ldB(SLOTP, cb, 1); \
  }
#define PSEG_END()                                   \
  __builtin_amdgcn_sched_barrier(0);                 \
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   \
  __builtin_amdgcn_s_barrier();                      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0);
#define PMMA(QI, QJ, FB)                                                                                         \
  __builtin_amdgcn_s_setprio(1);                                                                                 \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                               \
  _Pragma("unroll") for (int rb = 0; rb < 4; ++rb)                                                               \
  _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)                                                               \
    acc[QI * 4 + rb][QJ * 2 + cb] =                                                                              \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[cb][ks], fa[rb][ks], acc[QI * 4 + rb][QJ * 2 + cb], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);                                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                                             \
  asm volatile("" ::: "memory");                                                                                 \
  __builtin_amdgcn_s_barrier();                                                                                  \
  asm volatile("" ::: "memory");
#define PTILE(X, Y, PAR)                                             \
  {                                                                  \
    const char* buf = smem + (PAR) * BUF;                            \
    const char* nbuf = smem + (1 - (PAR)) * BUF;                     \
    ST_B1();                                                         \
    PREAD_A(buf + SA0)                                               \
    PSEG_END()                                                       \
    PMMA(0, 0, X)                                                    \
    ST_A1();                                                         \
    PREAD_B(Y, buf + SB1)                                            \
    PSEG_END()                                                       \
    PMMA(0, 1, Y)                                                    \
    ST_A0();                                                         \
    PREAD_A(buf + SA1)                                               \
    PSEG_END()                                                       \
    PMMA(1, 1, Y)                                                    \
    ST_B0();                                                         \
    PREAD_B(Y, nbuf + SB0)                                           \
    PSEG_END()                                                       \
    PMMA(1, 0, X)                                                    \
  }

  // epilogue of the current unit (same arithmetic and store pattern as gemm256_kernel), then clear the accumulators
  auto epilogue = [&]() __attribute__((always_inline)) {
    const int m0 = cu.m0, n0 = cu.n0;
    if (cu.part) {
      float* part = cu.part;
#pragma unroll
      for (int ri = 0; ri < 8; ++ri) {
        const int ml = (ri >> 2) * 128 + wr * 64 + (ri & 3) * 16 + li;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          const int nl = (ci >> 1) * 128 + wc * 32 + (ci & 1) * 16 + lg * 4;
          *(f32x4_t*)(part + ml * 256 + nl) = acc[ri][ci];
        }
      }
    } else {
      const bool vec_ok = ((p.ldc & 3) == 0) && (p.R == nullptr || (p.ldr & 3) == 0);
#pragma unroll
      for (int ri = 0; ri < 8; ++ri) {
        const int m = m0 + (ri >> 2) * 128 + wr * 64 + (ri & 3) * 16 + li;
        if (m >= p.M) continue;
#pragma unroll
        for (int ci = 0; ci < 4; ++ci) {
          const int n = n0 + (ci >> 1) * 128 + wc * 32 + (ci & 1) * 16 + lg * 4;
          if (n >= p.N) continue;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[ri][ci][r] * p.alpha;
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
              if (p.accumulate) o += *(const f32x4_t*)c;
              *(f32x4_t*)c = o;
            } else {
              u32x2_t o;
              o[0] = pack2bf(v[0], v[1]);
              o[1] = pack2bf(v[2], v[3]);
              *(u32x2_t*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
            }
          } else {
            for (int r = 0; r < 4 && n + r < p.N; ++r) {
              float xv = v[r];
              if (p.bias) xv += bf2f(p.bias[n + r]);
              if (p.R) xv += bf2f(p.R[(size_t)m * p.ldr + n + r]);
              if (p.out_fp32) {
                float* c = (float*)p.C + (size_t)m * p.ldc + n + r;
                *c = p.accumulate ? (*c + xv) : xv;
              } else {
                ((bf16_t*)p.C)[(size_t)m * p.ldc + n + r] = f2bf(xv);
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };

  // ---- prologue: K-tiles 0 (all four slots) and 1 (SA0, SB0) of the first unit
  ST_A0(); ST_B0(); ST_B1(); ST_A1(); ST_A0(); ST_B0();
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  PREAD_B(fb0, smem + SB0)
  if (wr == 1) __builtin_amdgcn_s_barrier();         // stagger wave row 1 by one barrier

  int ci_unit = 0, c_left = cu.nt;
  for (;;) {
    PTILE(fb0, fb1, 0)
    if (--c_left == 0) {
      epilogue();
      if (++ci_unit == n_units) break;
      cu = get_unit(ci_unit);
      c_left = cu.nt;
    }
    PTILE(fb1, fb0, 1)
    if (--c_left == 0) {
      epilogue();
      if (++ci_unit == n_units) break;
      cu = get_unit(ci_unit);
      c_left = cu.nt;
    }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();          // re-balance the stagger
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef PTILE
#undef PMMA
#undef PSEG_END
#undef PREAD_A
#undef PREAD_B
#undef PSTAGE
#undef PSLOT_SET
#undef ST_A0
#undef ST_A1
#undef ST_B0
#undef ST_B1
}

#endif  // MLA_EXPERIMENTAL_KERNELS

// Fix-up of the split-K tail: one thread per 4 consecutive outputs of a tail tile; sums the sk_split fp32 partials (fixed order:
// deterministic) and applies the same epilogue as the main kernel.
__global__ __launch_bounds__(256) void gemm256_fixup_kernel(GemmArgs p) {
  const int GROUP_M = MLA_GROUP_M;
  const int num_m = (p.M + 255) / 256, num_n = (p.N + 255) / 256;
  const int tile = blockIdx.x >> 6;                       // 64 blocks of 256 threads x 4 outputs per tile
  const int pid = p.sk_full + tile;
  const int in_group = GROUP_M * num_n;
  const int group_id = pid / in_group;
  const int first_m = group_id * GROUP_M;
  const int gsz = (num_m - first_m) < GROUP_M ? (num_m - first_m) : GROUP_M;
  const int pid_m = first_m + (pid % in_group) % gsz, pid_n = (pid % in_group) / gsz;
  const int e = ((blockIdx.x & 63) * 256 + threadIdx.x) * 4;
  const int ml = e >> 8, nl = e & 255;
  const int m = pid_m * 256 + ml, n = pid_n * 256 + nl;
  if (p.sq_out) {            // sum(C^2) partial of this block (fp32 vector path only: checked by the dispatcher)
    __shared__ float sq_s[16];
    float sq = 0.f;
    if (m < p.M && n < p.N) {
      const float* src = p.sk_ws + (size_t)tile * p.sk_split * 65536 + ml * 256 + nl;
      f32x4_t a = *(const f32x4_t*)src;
      for (int sl = 1; sl < p.sk_split; ++sl) a += *(const f32x4_t*)(src + (size_t)sl * 65536);
      f32x4_t o = a * p.alpha;
      if (p.bias) { const u32x2_t bb = *(const u32x2_t*)(p.bias + n); o[0] += bflo(bb[0]); o[1] += bfhi(bb[0]); o[2] += bflo(bb[1]); o[3] += bfhi(bb[1]); }
      if (p.R) { const u32x2_t rr = *(const u32x2_t*)(p.R + (size_t)m * p.ldr + n); o[0] += bflo(rr[0]); o[1] += bfhi(rr[0]); o[2] += bflo(rr[1]); o[3] += bfhi(rr[1]); }
      float* c = (float*)p.C + (size_t)m * p.ldc + n;
      if (p.accumulate) o += *(const f32x4_t*)c;
      *(f32x4_t*)c = o;
      sq = (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
    }
    sq = block_sum(sq, sq_s);
    if (threadIdx.x == 0) p.sq_out[p.sk_full + blockIdx.x] = sq;
    return;
  }
  if (m >= p.M || n >= p.N) return;
  const float* src = p.sk_ws + (size_t)tile * p.sk_split * 65536 + ml * 256 + nl;
  f32x4_t a = *(const f32x4_t*)src;
  for (int sl = 1; sl < p.sk_split; ++sl) a += *(const f32x4_t*)(src + (size_t)sl * 65536);
  float v[4] = {a[0] * p.alpha, a[1] * p.alpha, a[2] * p.alpha, a[3] * p.alpha};
  const bool vec_ok = ((p.ldc & 3) == 0) && (p.R == nullptr || (p.ldr & 3) == 0) && (n + 3 < p.N);
  for (int r = 0; r < 4 && n + r < p.N; ++r) {
    if (p.bias) v[r] += bf2f(p.bias[n + r]);
    if (p.R) v[r] += bf2f(p.R[(size_t)m * p.ldr + n + r]);
  }
  if (vec_ok) {
    if (p.out_fp32) {
      float* c = (float*)p.C + (size_t)m * p.ldc + n;
      f32x4_t o = {v[0], v[1], v[2], v[3]};
      if (p.accumulate) o += *(const f32x4_t*)c;
      *(f32x4_t*)c = o;
    } else {
      u32x2_t o;
      o[0] = pack2bf(v[0], v[1]);
      o[1] = pack2bf(v[2], v[3]);
      *(u32x2_t*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
      if (p.nrm_xg) {      // uniform (N % 256 == 0: a wave = the 256 columns of one row of the tile); same outputs as the main kernel's epilogue
        const u32x2_t gg = *(const u32x2_t*)(p.nrm_g + n);
        const float h0 = bflo(o[0]), h1 = bfhi(o[0]), h2 = bflo(o[1]), h3 = bfhi(o[1]);
        u32x2_t x;
        x[0] = pack2bf(h0 * bflo(gg[0]), h1 * bfhi(gg[0]));
        x[1] = pack2bf(h2 * bflo(gg[1]), h3 * bfhi(gg[1]));
        *(u32x2_t*)(p.nrm_xg + (size_t)m * p.ldc + n) = x;
        const float q = wave_sum((h0 * h0 + h1 * h1) + (h2 * h2 + h3 * h3));
        if ((threadIdx.x & 63) == 0) p.nrm_ss[(size_t)m * num_n + pid_n] = q;
      }
    }
  } else {
    for (int r = 0; r < 4 && n + r < p.N; ++r) {
      if (p.out_fp32) {
        float* c = (float*)p.C + (size_t)m * p.ldc + n + r;
        *c = p.accumulate ? (*c + v[r]) : v[r];
      } else {
        ((bf16_t*)p.C)[(size_t)m * p.ldc + n + r] = f2bf(v[r]);
      }
    }
  }
}

// Chooses the split of the tail: with T tiles on NCU compute units, T = q * NCU + r. The r tail tiles would occupy a whole
// round at r / NCU utilisation; cut into s K-slices they take ceil(r * s / NCU) / s of a round (+ the fix-up pass).
inline int choose_split(int tiles, int ncu, int nt, size_t ws_bytes, int* full_out) {
  const int r = tiles % ncu;
  *full_out = tiles - r;
  if (r == 0) return 1;
  int best = 1;
  double best_cost = 1.0;
  for (int s = 2; s <= 8; ++s) {
    if (nt / s < 8) break;                                            // keep >= 8 K-tiles per slice (pipeline fill / drain)
    if ((size_t)r * s * 65536 * sizeof(float) > ws_bytes) break;
    const double cost = (double)((r * s + ncu - 1) / ncu) / s + 0.04 + 0.01 * s;   // rounds; fix-up ~ 4-12 % of a round
    if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
  }
  return best;
}

// Number of CUs the launches plan their rounds for (split-K tail: tiles of the last, partially filled round of `ncu` workgroups).
// Default = the device's CU count; mla_gemm_cus(n) / MLA_GEMM_CUS=n plan for n (a multiple of 8) instead -- the knob for the multi-GPU
// run, where RCCL's kernels hold some CUs for most of the backward and a "round" is what is actually free.
int g_ncu_plan = -1;
int planned_cus(int device_cus) {
  if (g_ncu_plan < 0) {
    const char* e = getenv("MLA_GEMM_CUS");
    const int v = e ? atoi(e) : 0;
    g_ncu_plan = (v >= 8 && v % 8 == 0) ? v : 0;
  }
  return (g_ncu_plan > 0 && g_ncu_plan < device_cus) ? g_ncu_plan : device_cus;
}

// Main loop of the k-contiguous instantiations: 1 = hand-scheduled assembly (default), 0 = compiler-scheduled. Both give the same bits;
// MLA_GEMM_KLOOP=0 in the environment or mla_gemm_kloop(0) select the compiler's (A/B measurements, tests).
int g_kloop = -1;
int kloop_mode() {
  if (g_kloop < 0) {
    const char* e = getenv("MLA_GEMM_KLOOP");
    g_kloop = (e && e[0] == '0') ? 0 : 1;
  }
  return g_kloop;
}

template <int EPI>
void launch_asm(const GemmArgs& p, dim3 grid, size_t lds, hipStream_t stream) {
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)gemm256_kernel<0, 0, EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((gemm256_kernel<0, 0, EPI, true>), grid, dim3(512), lds, stream, p);
}

template <int AM, int BM_>
int launch256(const GemmArgs& p, hipStream_t stream, int persistent_grid) {
  if constexpr (AM != 0 || BM_ != 0) {   // reduction-major operands: the plain one-tile-per-workgroup launch only
    static bool attr_rm = false;
    if (!attr_rm) {
      hipFuncSetAttribute((const void*)gemm256_kernel<AM, BM_>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
      attr_rm = true;
    }
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (p.sk_split > 1) {
      const int tail = tiles - p.sk_full;
      hipLaunchKernelGGL((gemm256_kernel<AM, BM_>), dim3(p.sk_full + tail * p.sk_split), dim3(512), 2 * BUF, stream, p);
      hipLaunchKernelGGL(gemm256_fixup_kernel, dim3(tail * 64), dim3(256), 0, stream, p);
    } else {
      hipLaunchKernelGGL((gemm256_kernel<AM, BM_>), dim3(tiles), dim3(512), 2 * BUF, stream, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      mla_set_error("gemm256 launch failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    return 0;
  } else {
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm256_kernel<AM, BM_>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF + 1024);
    attr_set = true;
  }
  const int num_m = (p.M + 255) / 256, num_n = (p.N + 255) / 256;
  // hand-scheduled main loop (template parameter ASM): an even number of K-tiles and the whole-row epilogue
  const bool fast = (p.N & 7) == 0 && (p.ldc & 7) == 0 && (((uintptr_t)p.C) & 15) == 0 &&
                    (p.R == nullptr || ((p.ldr & 7) == 0 && (((uintptr_t)p.R) & 15) == 0)) && (p.bias == nullptr || (((uintptr_t)p.bias) & 15) == 0);
  if (kloop_mode() && (p.K % 128) == 0 && fast && persistent_grid <= 0) {
    if (p.sk_split > 1) {
      const int tail = num_m * num_n - p.sk_full;
      launch_asm<0>(p, dim3(p.sk_full + tail * p.sk_split), 2 * BUF + 1024, stream);
      hipLaunchKernelGGL(gemm256_fixup_kernel, dim3(tail * 64), dim3(256), 0, stream, p);
    } else if (p.sw_gu) {
      launch_asm<2>(p, dim3(num_m * num_n), 2 * BUF + 32768, stream);
    } else if (p.sf_I) {
      launch_asm<1>(p, dim3(num_m * num_n), 2 * BUF + 1024, stream);       // + 1 KiB: row scales of a folded RMSNorm (rs_*)
    } else {
      launch_asm<0>(p, dim3(num_m * num_n), 2 * BUF + 1024, stream);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      mla_set_error("gemm256 launch failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    return 0;
  }
#ifdef MLA_EXPERIMENTAL_KERNELS
  if (AM == 0 && BM_ == 0 && persistent_grid > 0) {
    static bool attr_p = false;
    if (!attr_p) {
      hipFuncSetAttribute((const void*)gemm256p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
      attr_p = true;
    }
    hipLaunchKernelGGL(gemm256p_kernel, dim3(persistent_grid), dim3(512), 2 * BUF, stream, p);
    if (p.sk_split > 1) hipLaunchKernelGGL(gemm256_fixup_kernel, dim3((num_m * num_n - p.sk_full) * 64), dim3(256), 0, stream, p);
  } else
#endif
  if (p.sk_split > 1) {
    const int tail = num_m * num_n - p.sk_full;
    hipLaunchKernelGGL((gemm256_kernel<AM, BM_>), dim3(p.sk_full + tail * p.sk_split), dim3(512), 2 * BUF + 1024, stream, p);
    hipLaunchKernelGGL(gemm256_fixup_kernel, dim3(tail * 64), dim3(256), 0, stream, p);
  } else if (p.sw_gu) {
    static bool attr_sw = false;      // the fused SwiGLU-backward epilogue stages a second transposed strip in 32 KiB beyond the ring
    if (!attr_sw) {
      hipFuncSetAttribute((const void*)gemm256_kernel<AM, BM_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF + 32768);
      attr_sw = true;
    }
    hipLaunchKernelGGL((gemm256_kernel<AM, BM_, 2>), dim3(num_m * num_n), dim3(512), 2 * BUF + 32768, stream, p);
  } else if (p.sf_I) {
    static bool attr_sf = false;
    if (!attr_sf) {
      hipFuncSetAttribute((const void*)gemm256_kernel<AM, BM_, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF + 1024);
      attr_sf = true;
    }
    hipLaunchKernelGGL((gemm256_kernel<AM, BM_, 1>), dim3(num_m * num_n), dim3(512), 2 * BUF + 1024, stream, p);
  } else {
    hipLaunchKernelGGL((gemm256_kernel<AM, BM_>), dim3(num_m * num_n), dim3(512), 2 * BUF + 1024, stream, p);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mla_set_error("gemm256 launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
  }
}

}  // namespace

int mla_gemm256_dispatch(const void* args, int a_mode, int b_mode, size_t ws_bytes, hipStream_t stream, int* sq_slots = nullptr);

// Selects the main loop of gemm256's k-contiguous instantiations: 1 = hand-scheduled assembly (default), 0 = compiler-scheduled, any other
// value only queries. Returns the mode in force afterwards. Results do not depend on it (same accumulation order); it exists for A/B runs.
extern "C" int mla_gemm_kloop(int mode) {
  if (mode == 0 || mode == 1) g_kloop = mode;
  return kloop_mode();
}

// Fused QKV projection + rotary embedding: C[M, N] = A[M, K] B[N, K]^T (bf16), columns [0, rope_cols) rotated per head of 128 with
// position = row % S. Replaces hip.gemm + mla_rope_inplace on the packed q|k|v buffer (LlamaAttention.forward :351-361).
extern "C" int mla_gemm_qkv_rope(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                 const float* rope_cos, const float* rope_sin, int S, int rope_cols, hipStream_t stream) {
  MLA_CHECK_ARG(A && B && C && rope_cos && rope_sin, "mla_gemm_qkv_rope: null pointer");
  MLA_CHECK_ARG(M >= 256 && N >= 256 && K > 0 && K % 64 == 0 && N % 8 == 0, "mla_gemm_qkv_rope: needs M, N >= 256, K %% 64 == 0, N %% 8 == 0");
  MLA_CHECK_ARG(S > 0 && rope_cols > 0 && rope_cols % 256 == 0 && rope_cols <= N, "mla_gemm_qkv_rope: rope_cols must be a multiple of 256 (two heads of 128) and <= N");
  MLA_CHECK_ARG(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "mla_gemm_qkv_rope: leading dimensions must be multiples of 8");
  MLA_CHECK_ARG(((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)rope_cos) | ((uintptr_t)rope_sin)) & 15) == 0,
                "mla_gemm_qkv_rope: 16-B alignment required");
  GemmArgs p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.alpha = 1.f;
  p.rope_cos = rope_cos; p.rope_sin = rope_sin; p.rope_S = S; p.rope_cols = rope_cols;
  return mla_gemm256_dispatch(&p, 0, 0, 0, stream);     // no split-K tail (its fix-up pass has no rotary epilogue)
}

// Fused gate|up projection + SwiGLU: gu[M, 2I] = x[M, K] wgu[2I, K]^T (stored, the backward needs it), act = silu(gate) * up [M, I] and
// (optional) its transpose [I, ldt]. Replaces hip.gemm + mla_swiglu_fwd_dual (LlamaMLP.forward modeling_llama.py:240); bit-identical.
extern "C" int mla_gemm_gateup_swiglu(const void* x, const void* wgu, void* gu, void* act, void* actT, int M, int I, int K, int lda,
                                      int ldb, long long ldt, hipStream_t stream) {
  MLA_CHECK_ARG(x && wgu && (act || actT) && (gu || act), "mla_gemm_gateup_swiglu: null pointer (act may be NULL only when actT is given, gu only when act is)");
  MLA_CHECK_ARG(M >= 256 && I >= 128 && I % 128 == 0 && K > 0 && K % 64 == 0, "mla_gemm_gateup_swiglu: needs M >= 256, I %% 128 == 0, K %% 64 == 0");
  MLA_CHECK_ARG(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0, "mla_gemm_gateup_swiglu: leading dimensions must be multiples of 8");
  MLA_CHECK_ARG(actT == nullptr || (M % 8 == 0 && ldt >= M && ldt % 8 == 0), "mla_gemm_gateup_swiglu: transposed output needs M %% 8 == 0, ldt >= M, ldt %% 8 == 0");
  MLA_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)wgu) | ((uintptr_t)gu) | ((uintptr_t)act) | ((uintptr_t)actT)) & 15) == 0,
                "mla_gemm_gateup_swiglu: 16-B alignment required");
  GemmArgs p{};
  p.A = (const bf16_t*)x; p.B = (const bf16_t*)wgu; p.C = gu; p.M = M; p.N = 2 * I; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = 2 * I;
  p.alpha = 1.f;
  p.sf_I = I; p.sf_act = (bf16_t*)act; p.sf_actT = (bf16_t*)actT; p.sf_ldt = ldt;
  return mla_gemm256_dispatch(&p, 0, 0, 0, stream);     // no split-K tail (its fix-up pass has no SwiGLU epilogue)
}

// Fused d(act) GEMM + SwiGLU backward: d(act) = dy[M, K] wT[I, K]^T stays on the chip; outputs d(gate|up) [M, 2I] and its transpose
// [2I, ldt]. Replaces hip.gemm + mla_swiglu_bwd_t in the MLP backward (autograd of modeling_llama.py:240); bit-identical to them.
extern "C" int mla_gemm_dact_swiglu_bwd(const void* dy, const void* wT, const void* gu, void* dgu, void* dguT, int M, int I, int K,
                                        int lda, int ldb, long long ldt, hipStream_t stream) {
  MLA_CHECK_ARG(dy && wT && gu && dgu && dguT, "mla_gemm_dact_swiglu_bwd: null pointer");
  MLA_CHECK_ARG(M >= 256 && I >= 256 && K > 0 && K % 64 == 0 && I % 8 == 0 && M % 8 == 0,
                "mla_gemm_dact_swiglu_bwd: needs M, I >= 256, K %% 64 == 0, I %% 8 == 0, M %% 8 == 0");
  MLA_CHECK_ARG(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 && ldt >= M && ldt % 8 == 0,
                "mla_gemm_dact_swiglu_bwd: leading dimensions must be multiples of 8 (ldt >= M)");
  MLA_CHECK_ARG(((((uintptr_t)dy) | ((uintptr_t)wT) | ((uintptr_t)gu) | ((uintptr_t)dgu) | ((uintptr_t)dguT)) & 15) == 0,
                "mla_gemm_dact_swiglu_bwd: 16-B alignment required");
  GemmArgs p{};
  p.A = (const bf16_t*)dy; p.B = (const bf16_t*)wT; p.C = dgu /* never written: the epilogue returns before the plain stores */;
  p.M = M; p.N = I; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = 2 * I; p.alpha = 1.f;
  p.sw_gu = (const bf16_t*)gu; p.sw_dgu = (bf16_t*)dgu; p.sw_dguT = (bf16_t*)dguT; p.sw_I = I; p.sw_ldt = ldt;
  return mla_gemm256_dispatch(&p, 0, 0, 0, stream);     // no split-K tail (its fix-up pass has no SwiGLU epilogue)
}

// ---- RMSNorm folded into the projections (round 6; the kernel north_star names: fused RMSNorm + RoPE + QKV). LlamaRMSNorm
// (modeling_llama.py:76-90) computes y = g * bf16(x * rstd) and the projection (:351-353, :240) y W^T. The row scale commutes with the
// product: y W^T = rstd (.) ((x * g) W^T). So the GEMM that PRODUCES x (o_proj / down_proj + residual) also leaves x * g and the
// per-tile partials of sum(x^2) (mla_gemm_res_norm), and the projection that CONSUMES them multiplies its fp32 accumulator rows by rstd
// before the single rounding to bf16 that the fused RoPE / SwiGLU epilogues start from (the _rs forms below). The stand-alone norm pass
// (read x, write y: 2 x 2 B per element) disappears; what is added is the x * g store (2 B per element) in the producer's epilogue.
// Rounding points: x * g is rounded once (the reference rounds x * rstd and then g * that), the projection is rounded once after the
// fp32 row scale (the reference: once) -- one rounding fewer on the way, none more.
extern "C" int mla_gemm_res_norm(const void* A, const void* B, void* C, const void* R, const void* g, void* xg, float* ss, int M, int N,
                                 int K, int lda, int ldb, int ldc, int ldr, float* workspace, size_t workspace_bytes, hipStream_t stream) {
  MLA_CHECK_ARG(A && B && C && R && g && xg && ss, "mla_gemm_res_norm: null pointer");
  MLA_CHECK_ARG(M >= 256 && N >= 256 && N % 256 == 0 && K > 0 && K % 64 == 0, "mla_gemm_res_norm: needs M >= 256, N %% 256 == 0, K %% 64 == 0");
  MLA_CHECK_ARG(lda >= K && ldb >= K && ldc >= N && ldr >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && ldr % 8 == 0,
                "mla_gemm_res_norm: leading dimensions must be multiples of 8");
  MLA_CHECK_ARG(((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)R) | ((uintptr_t)g) | ((uintptr_t)xg) | ((uintptr_t)ss) |
                  ((uintptr_t)workspace)) & 15) == 0, "mla_gemm_res_norm: 16-B alignment required");
  GemmArgs p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.R = (const bf16_t*)R; p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.alpha = 1.f;
  p.sk_ws = workspace;
  p.nrm_g = (const bf16_t*)g; p.nrm_xg = (bf16_t*)xg; p.nrm_ss = ss;      // xg has C's leading dimension; ss is [M, N / 256]
  return mla_gemm256_dispatch(&p, 0, 0, workspace ? workspace_bytes : 0, stream);
}

static int rs_args(GemmArgs& p, const float* ss, int parts, float eps, float* rstd, const char* who) {
  MLA_CHECK_ARG(rstd != nullptr && (ss == nullptr || parts > 0), "%s: null rstd / bad partial count", who);
  p.rs_ss = ss; p.rs_parts = parts; p.rs_eps = eps; p.rs_rstd = rstd;
  return 0;
}

// mla_gemm_qkv_rope on A = x * g with the row scale of the folded RMSNorm: ss [M, parts] partials of sum(x^2) (rstd is computed here and
// stored to rstd [M]) or ss == NULL (rstd [M] is read). Fused RMSNorm + QKV + RoPE in one launch.
extern "C" int mla_gemm_qkv_rope_rs(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                                    const float* rope_cos, const float* rope_sin, int S, int rope_cols, const float* ss, int parts,
                                    float eps, float* rstd, hipStream_t stream) {
  MLA_CHECK_ARG(A && B && C && rope_cos && rope_sin, "mla_gemm_qkv_rope_rs: null pointer");
  MLA_CHECK_ARG(M >= 256 && N >= 256 && K > 0 && K % 64 == 0 && N % 8 == 0, "mla_gemm_qkv_rope_rs: needs M, N >= 256, K %% 64 == 0, N %% 8 == 0");
  MLA_CHECK_ARG(S > 0 && rope_cols > 0 && rope_cols % 256 == 0 && rope_cols <= N, "mla_gemm_qkv_rope_rs: rope_cols must be a multiple of 256 (two heads of 128) and <= N");
  MLA_CHECK_ARG(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "mla_gemm_qkv_rope_rs: leading dimensions must be multiples of 8");
  MLA_CHECK_ARG(((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)rope_cos) | ((uintptr_t)rope_sin)) & 15) == 0,
                "mla_gemm_qkv_rope_rs: 16-B alignment required");
  GemmArgs p{};
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.alpha = 1.f;
  p.rope_cos = rope_cos; p.rope_sin = rope_sin; p.rope_S = S; p.rope_cols = rope_cols;
  if (int rc = rs_args(p, ss, parts, eps, rstd, "mla_gemm_qkv_rope_rs")) return rc;
  return mla_gemm256_dispatch(&p, 0, 0, 0, stream);
}

// mla_gemm_gateup_swiglu on A = x * g with the row scale of the folded RMSNorm (see mla_gemm_qkv_rope_rs).
extern "C" int mla_gemm_gateup_swiglu_rs(const void* x, const void* wgu, void* gu, void* act, void* actT, int M, int I, int K, int lda,
                                         int ldb, long long ldt, const float* ss, int parts, float eps, float* rstd, hipStream_t stream) {
  MLA_CHECK_ARG(x && wgu && (act || actT) && (gu || act), "mla_gemm_gateup_swiglu_rs: null pointer (act may be NULL only when actT is given, gu only when act is)");
  MLA_CHECK_ARG(M >= 256 && I >= 128 && I % 128 == 0 && K > 0 && K % 64 == 0, "mla_gemm_gateup_swiglu_rs: needs M >= 256, I %% 128 == 0, K %% 64 == 0");
  MLA_CHECK_ARG(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0, "mla_gemm_gateup_swiglu_rs: leading dimensions must be multiples of 8");
  MLA_CHECK_ARG(actT == nullptr || (M % 8 == 0 && ldt >= M && ldt % 8 == 0), "mla_gemm_gateup_swiglu_rs: transposed output needs M %% 8 == 0, ldt >= M, ldt %% 8 == 0");
  MLA_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)wgu) | ((uintptr_t)gu) | ((uintptr_t)act) | ((uintptr_t)actT)) & 15) == 0,
                "mla_gemm_gateup_swiglu_rs: 16-B alignment required");
  GemmArgs p{};
  p.A = (const bf16_t*)x; p.B = (const bf16_t*)wgu; p.C = gu; p.M = M; p.N = 2 * I; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = 2 * I;
  p.alpha = 1.f;
  p.sf_I = I; p.sf_act = (bf16_t*)act; p.sf_actT = (bf16_t*)actT; p.sf_ldt = ldt;
  if (int rc = rs_args(p, ss, parts, eps, rstd, "mla_gemm_gateup_swiglu_rs")) return rc;
  return mla_gemm256_dispatch(&p, 0, 0, 0, stream);
}

// called by mla_gemm_bf16 (gemm.hip) for k-contiguous operands with M, N >= 256 and K % 64 == 0. Only the <0,0>
// instantiation is built: the reduction-major (ds_read_b64_tr_b16) variants of this schedule are slower than gemm128's.
// number of sum-of-squares partials a k-contiguous fp32-output launch of this shape writes (whole tiles + 64 per split tile): lets the
// caller reserve exactly that many floats in a shared buffer and sum the whole buffer in one launch afterwards
// n >= 8 (a multiple of 8): plan the split-K tails for n CUs; n == 0: back to the device's CU count; n < 0: query. Returns the value in
// force (0 = device count), -1 for a bad argument. Set it before the work starts (not synchronised with concurrent launches).
extern "C" int mla_gemm_cus(int n) {
  if (n >= 0) {
    if (n != 0 && (n < 8 || n % 8 != 0 || n > 1024)) return -1;
    g_ncu_plan = n;
  } else if (g_ncu_plan < 0) {
    (void)planned_cus(1 << 20);
  }
  return g_ncu_plan;
}
extern "C" int mla_gemm_sq_slots(int M, int N, int K, size_t workspace_bytes) {
  if (M < 256 || N < 256 || K <= 0 || (K % 64) != 0 || (N % 8) != 0) return -1;
  int ncu = 0, dev = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (ncu <= 0) ncu = 256;
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  if (workspace_bytes) {
    int full = 0;
    const int s = choose_split(tiles, planned_cus(ncu), K / 64, workspace_bytes, &full);
    if (s > 1 && full % 8 == 0) return full + (tiles - full) * 64;
  }
  return tiles;
}

int mla_gemm256_dispatch(const void* args, int a_mode, int b_mode, size_t ws_bytes, hipStream_t stream, int* sq_slots) {
  GemmArgs p = *(const GemmArgs*)args;
  if ((a_mode != 0 || b_mode != 0) && (p.sf_I || p.sw_gu || p.rope_cos)) {
    mla_set_error("gemm256: the fused epilogues take k-contiguous operands only");
    return -1;
  }
  p.sk_split = 1;
  p.sk_full = 0;
  if (p.nrm_xg || p.rs_rstd) {     // folded RMSNorm: whole-row epilogues of the k-contiguous kernel only (the entry points check the shapes)
    const bool fast = (p.N & 7) == 0 && (p.ldc & 7) == 0 && (((uintptr_t)p.C) & 15) == 0 &&
                      (p.R == nullptr || ((p.ldr & 7) == 0 && (((uintptr_t)p.R) & 15) == 0)) && p.bias == nullptr && !p.out_fp32;
    const bool ok = fast && a_mode == 0 && b_mode == 0 && p.alpha == 1.f &&
                    (p.nrm_xg ? (p.R != nullptr && p.N % 256 == 0 && !p.rs_rstd && !p.sf_I && !p.sw_gu && !p.rope_cos) : (p.R == nullptr && !p.sw_gu));
    if (!ok) {
      mla_set_error("gemm256: the folded-RMSNorm forms need bf16 k-contiguous operands, alpha 1 and the whole-row epilogue");
      return -1;
    }
  }
  static int ncu = 0, persist = -1;
  if (!ncu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu <= 0) ncu = 256;
    // opt-in (MLA_GEMM_PERSIST=1): bit-identical to the one-workgroup-per-tile launch and measured -1.5 ... +0.3 % against it on the
    // twelve 7B shapes: hiding the per-tile pipeline fill is paid back by the unit bookkeeping in the hot loop
    const char* e = getenv("MLA_GEMM_PERSIST");
    persist = (e && e[0] == '1') ? 1 : 0;
  }
#ifdef MLA_EXPERIMENTAL_KERNELS
  if (p.sf_I || p.sw_gu) {
    static int st_mode = -1, st_ticks = 0;
    if (st_mode < 0) {
      const char* e = getenv("MLA_GEMM_STAGGER");      // "<mode>:<ticks of 10 ns>", e.g. 1:4500
      st_mode = 0;
      if (e && sscanf(e, "%d:%d", &st_mode, &st_ticks) != 2) st_mode = 0;
    }
    p.stagger_mode = st_mode;
    p.stagger_ticks = st_ticks;
  }
#endif
  const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  if (p.sq_out) {   // sum-of-squares partials: fp32 output through the whole-row (fast) epilogue, plain instantiation only
    const bool ok = p.out_fp32 && a_mode == 0 && b_mode == 0 && !p.sf_I && !p.sw_gu && !p.rope_cos && (p.N & 7) == 0 && (p.ldc & 7) == 0 &&
                    (((uintptr_t)p.C) & 15) == 0 && (p.R == nullptr || ((p.ldr & 7) == 0 && (((uintptr_t)p.R) & 15) == 0)) &&
                    (p.bias == nullptr || (((uintptr_t)p.bias) & 15) == 0);
    if (!ok) {
      mla_set_error("gemm256: sum-of-squares partials need an fp32, 8-column-aligned output of the plain kernel");
      return -1;
    }
  }
  if (p.sk_ws && ws_bytes) {
    int full = 0;
    const int s = choose_split(tiles, planned_cus(ncu), p.K / 64, ws_bytes, &full);
    if (s > 1 && full % 8 == 0) { p.sk_split = s; p.sk_full = full; }
  }
  // persistent walk when there is more than one round of tiles and the operands fit 31-bit byte offsets (buffer addressing)
  const size_t bytesA = ((size_t)(p.M - 1) * p.lda + p.K) * 2, bytesB = ((size_t)(p.N - 1) * p.ldb + p.K) * 2;
  const int grid = ncu & ~7;
#ifndef MLA_EXPERIMENTAL_KERNELS
  if (p.debug & 0x100) { mla_set_error("gemm256: the persistent experiment kernel is not in this build (MLA_EXPERIMENTAL=1)"); return -1; }
  persist = 0;
#endif
  const bool use_p = (persist || (p.debug & 0x100)) && (p.debug & 0x80) == 0 && tiles > grid && bytesA < 0x7fffffffULL && bytesB < 0x7fffffffULL &&
                     p.sf_I == 0 && p.sw_gu == nullptr && p.rope_cos == nullptr && p.sq_out == nullptr && p.nrm_xg == nullptr && p.rs_rstd == nullptr;     // the persistent walk has the plain epilogue only
  if (sq_slots) *sq_slots = p.sk_split > 1 ? p.sk_full + (tiles - p.sk_full) * 64 : tiles;
  // reduction-major operands ([K, rows] storage, fragments gathered with ds_read_b64_tr_b16): plain epilogue, split-K tail allowed
  if (a_mode == 0 && b_mode == 1) return launch256<0, 1>(p, stream, 0);
  if (a_mode == 1 && b_mode == 0) return launch256<1, 0>(p, stream, 0);
  if (a_mode == 1 && b_mode == 1) return launch256<1, 1>(p, stream, 0);
  return launch256<0, 0>(p, stream, use_p ? grid : 0);
}
