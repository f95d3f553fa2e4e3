// Stand-in for a collective's kernel on the side stream (round 4, VERDICT r3 next #3c): the 8-GPU run will have RCCL's reduce-scatter
// kernels sharing the chip with the backward GEMMs for most of every backward (23.6 GB of fp32 gradients out per GPU and step at
// N = 8), and no 8-GPU node is available to the builder. mla_side_traffic reproduces what such a kernel does to the rest of the chip
// -- `blocks` workgroups resident for the duration, each streaming a slice of `a` and `b` (16 B per lane) into `out = a + b`, with a
// sleep between chunks so that the slice lasts as long as a link-bound ring step would -- so that the step's sensitivity to k
// occupied CUs can be measured on one GPU (tools/contention_rehearsal.py -> profiles/r4_contention.txt).
//   lds_bytes > 0: the workgroup allocates that much LDS. 160 KiB makes it the only resident workgroup of its CU (a gemm256
//   workgroup needs 160 KiB itself), i.e. the CU is TAKEN from the GEMM; 0 leaves the placement to the dispatcher (sixteen waves of < 32 registers).
#include "common.h"

namespace {

__global__ __launch_bounds__(1024) void side_traffic_kernel(const f32x4_t* __restrict__ a, const f32x4_t* __restrict__ b, f32x4_t* __restrict__ out,
                                                           long long n4, int sleep_ticks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  (void)smem;
  const long long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long long lo = per * blockIdx.x, hi = (lo + per) < n4 ? (lo + per) : n4;
  // chunks of 1024 threads x 4 x 16 B = 64 KiB per operand, all 8 loads of a thread in flight before the first use (sixteen waves:
  // a 256-thread workgroup moved 16 GB/s whatever its unrolling -- latency-bound, far below what a collective's kernel does per channel)
  for (long long base = lo; base < hi; base += 4096) {
    f32x4_t va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = base + u * 1024 + threadIdx.x;
      if (i < hi) { va[u] = __builtin_nontemporal_load(a + i); vb[u] = __builtin_nontemporal_load(b + i); }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = base + u * 1024 + threadIdx.x;
      if (i < hi) __builtin_nontemporal_store(va[u] + vb[u], out + i);
    }
    for (int t = 0; t < sleep_ticks; ++t) __builtin_amdgcn_s_sleep(127);     // 127 x 64 clocks per tick
  }
}

}  // namespace

extern "C" int mla_side_traffic(const float* a, const float* b, float* out, long long n, int blocks, int lds_bytes, int sleep_ticks,
                                hipStream_t stream) {
  MLA_CHECK_ARG(a && b && out && n > 0 && (n & 3) == 0, "mla_side_traffic: null pointer or n %% 4 != 0");
  MLA_CHECK_ARG(((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15) == 0, "mla_side_traffic: 16-B alignment required");
  MLA_CHECK_ARG(blocks > 0 && blocks <= 1024 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && sleep_ticks >= 0, "mla_side_traffic: bad launch shape");
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)side_traffic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
  hipLaunchKernelGGL(side_traffic_kernel, dim3(blocks), dim3(1024), (size_t)lds_bytes, stream, (const f32x4_t*)a, (const f32x4_t*)b, (f32x4_t*)out,
                     n / 4, sleep_ticks);
  MLA_LAUNCH_CHECK();
}
