// Does de-synchronising the compute units help when a kernel alternates an MFMA phase and a store burst (the fused GEMM epilogues)?
// Every workgroup (512 threads, one per CU) runs ROUNDS x { MFMA loop on registers (~main loop of one tile), store `bytes` (the tile's
// epilogue) }. stagger 0: all CUs in phase (what equal tiles give); 1: odd XCDs start with HALF an MFMA phase, so their bursts fall into
// the even XCDs' MFMA phases (same total work per CU); 2: every other workgroup instead (breaks an XCD's lockstep too).
// Prints time per round. Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o phase_overlap phase_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ int xcc_id() { int v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15; }

__global__ __launch_bounds__(512, 2) void k(u32x4* __restrict__ dst, long long per_wg16, int rounds, int mfma_iters, int stagger, int do_store) {
  const long long base = (long long)blockIdx.x * per_wg16;
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.5f + 0.01f * i); }
  const bool late = stagger == 1 ? (xcc_id() & 1) : stagger == 2 ? (blockIdx.x >> 3) & 1 : false;
  for (int r = 0; r < rounds; ++r) {
    const int it = (r == 0 && late) ? mfma_iters / 2 : mfma_iters;
    for (int i = 0; i < it; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
      { u32x4 t = __builtin_bit_cast(u32x4, a); t[i & 3] = t[i & 3] * 1664525u + 1013904223u; t[(i + 1) & 3] &= 0xbf7fbf7fu;   // keep the operands changing
        a = __builtin_bit_cast(bf16x8, t); }
    }
    if (do_store) {
      u32x4 v = {__float_as_uint(acc[0][0]), __float_as_uint(acc[5][1]), __float_as_uint(acc[9][2]), (unsigned)r};
      for (long long i = threadIdx.x; i < per_wg16; i += 512) dst[base + i] = v;
    }
  }
  if (acc[3][3] == 12345.f) dst[base] = u32x4{1, 2, 3, 4};
}

int main() {
  const long long per_wg = 640 << 10;
  const int rounds = 24;
  u32x4* d;
  hipMalloc(&d, 256 * per_wg);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int iters : {400, 200, 100}) {
    for (int mode = 0; mode < 4; ++mode) {     // 3 = no stores at all (the MFMA phases alone)
      const int stagger = mode == 3 ? 0 : mode, st = mode != 3;
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, per_wg / 16, rounds, iters, stagger, st);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, per_wg / 16, rounds, iters, stagger, st);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mfma_iters %4d  %-28s %8.1f us per round\n", iters, mode == 0 ? "lockstep" : mode == 1 ? "odd XCDs half a phase late" : mode == 2 ? "every other WG late" : "no stores (MFMA only)",
             ms / 3 * 1e3f / rounds);
    }
  }
  return 0;
}
