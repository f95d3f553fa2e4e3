// Error plumbing + ABI query for libmla_hip.so (see include/mla_hip.h).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void mla_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mla_last_error(void) { return g_err; }

// sha256[:16] over the sources of the gemm256 kernel family (gemm256.hip, its generated main-loop .inc files, gemm_args.h, common.h),
// stamped by build.sh: profiles/*_hbm_traffic.json records the id of the kernel the counters were collected on, and bench.py flags
// `traffic_stale` when the loaded library's id differs.
#ifndef MLA_GEMM_SRC_ID
#define MLA_GEMM_SRC_ID "unknown"
#endif
extern "C" const char* mla_gemm_source_id(void) { return MLA_GEMM_SRC_ID; }

// what: 0 = ABI version, 1 = compiled gfx arch number (950), 2 = wavefront size the kernels assume, 3 = 1 when the opt-in experiment
// kernels (assembly GEMM main loops, persistent GEMM walk) were compiled in (build.sh MLA_EXPERIMENTAL=1), else 0
extern "C" int mla_query(int what) {
  switch (what) {
    case 0: return 1;
    case 1: return 950;
    case 2: return 64;
#ifdef MLA_EXPERIMENTAL_KERNELS
    case 3: return 1;
#else
    case 3: return 0;
#endif
    default: return -1;
  }
}

// ---- hardware-assumption self tests (run by tests/test_hip_selftest.py on the GPU box) ----
// out_tr[lane*4 + j] = element returned to `lane` as its j-th b16 by ds_read_b64_tr_b16 when LDS holds
// lds[e] = e (u16) and lane l passes byte address l*8.  DESIGN.md states the mapping the kernels rely on:
// out_tr[l*4+j] == (l>>4)*64 + j*16 + (l&15).
// out_glds[i] = u32 word i of the LDS image after one global_load_lds_dwordx4 where lane l sourced src + l*16 bytes:
// expected out_glds[i] == src_words[i] (lane-linear destination).
__global__ void selftest_kernel(const uint32_t* __restrict__ src, int* __restrict__ out_tr, uint32_t* __restrict__ out_glds) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* l16 = (uint16_t*)smem;
  const int lane = threadIdx.x;
  for (int e = lane; e < 1024; e += 64) l16[e] = (uint16_t)e;
  __syncthreads();
  const short4_t v = lds_tr16_b64(smem + lane * 8);
  for (int j = 0; j < 4; ++j) out_tr[lane * 4 + j] = (int)(uint16_t)v[j];
  __syncthreads();
  glds16((const char*)src + lane * 16, smem + 4096);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const uint32_t* l32 = (const uint32_t*)(smem + 4096);
  for (int i = lane; i < 256; i += 64) out_glds[i] = l32[i];
}

extern "C" int mla_selftest(const void* src_1k, int* out_tr_256, void* out_glds_1k, hipStream_t stream) {
  MLA_CHECK_ARG(src_1k && out_tr_256 && out_glds_1k, "mla_selftest: null pointer");
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 8192, stream, (const uint32_t*)src_1k, out_tr_256, (uint32_t*)out_glds_1k);
  MLA_LAUNCH_CHECK();
}

// ---- dispatch probe (round 6): the two things attn_bwd_merged_kernel assumes about the hardware queue and HIP does not promise --
//   (1) workgroup id L runs on XCD L & 7 (its per-head counters and the L2 reuse of q, k, v, dO are laid out for that), 8 XCDs;
//   (2) workgroups start in id order per XCD: a consumer of a head (higher id) never holds a CU slot while one of its producers (lower
//       id, same XCD) has not been given one -- the condition under which its wait cannot dead-lock.
// `blocks` workgroups shaped like that kernel's (256 threads, 2 per CU by LDS) each take a ticket from out[0] when they start, stay
// resident for ~hold_us (so the grid needs many dispatch rounds), and record out[1 + 2 L] = ticket, out[2 + 2 L] = XCC_ID. The caller
// (mla_amd/hip.py: dispatch_probe) checks (1) literally and (2) as "no workgroup took its ticket more than two residency rounds (2 x 64
// slots per XCD) out of id order", and hands head counters to mla_attn_bwd only on a device where both hold -- the two-launch form otherwise.
__global__ __launch_bounds__(256, 2) void dispatch_probe_kernel(int* __restrict__ out, int hold_ticks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  (void)smem;
  if (threadIdx.x == 0) {
    const int ticket = __hip_atomic_fetch_add(out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)hold_ticks) __builtin_amdgcn_s_sleep(16);
    out[1 + 2 * blockIdx.x] = ticket;
    out[2 + 2 * blockIdx.x] = xcc & 15;
  }
}

extern "C" int mla_dispatch_probe(int* out, int blocks, int hold_us, hipStream_t stream) {
  MLA_CHECK_ARG(out && blocks > 0 && blocks <= (1 << 20) && hold_us >= 0 && hold_us <= 10000, "mla_dispatch_probe: null pointer or bad shape");
  static bool attr = false;
  constexpr int LDS = 80 * 1024;                    // two workgroups per CU, like the attention backward
  if (!attr) { (void)hipFuncSetAttribute((const void*)dispatch_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
  hipLaunchKernelGGL(dispatch_probe_kernel, dim3(blocks), dim3(256), LDS, stream, out, hold_us * 100);   // s_memrealtime: 100 MHz
  MLA_LAUNCH_CHECK();
}
