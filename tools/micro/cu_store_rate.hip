// What ONE compute unit can move: a 512-thread workgroup per CU streams `bytes` per workgroup as 16-B accesses per lane, rows of 512 B
// per half-wave (the shape of the fused GEMM epilogues' row stores). Modes: 0 = store only, 1 = load only, 2 = load + store (copy).
// Grid = number of workgroups (256 = every CU busy: the chip's memory system is shared; 16 = two CUs per XCD: what a CU can do alone).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o cu_store_rate cu_store_rate.hip ; run: ./cu_store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, int NT>
__global__ __launch_bounds__(512) void k(u32x4* __restrict__ dst, const u32x4* __restrict__ src, long long per_wg16, int reps) {
  const long long base = (long long)blockIdx.x * per_wg16;
  u32x4 acc = {threadIdx.x, 1u, 2u, 3u};
  for (int r = 0; r < reps; ++r)
    for (long long i = threadIdx.x; i < per_wg16; i += 512) {
      u32x4 v = acc;
      if (MODE != 0) v = NT ? __builtin_nontemporal_load(src + base + i) : src[base + i];
      if (MODE == 1) acc ^= v;
      if (MODE != 1) { if (NT) __builtin_nontemporal_store(v, dst + base + i); else dst[base + i] = v; }
    }
  if (MODE == 1 && acc[0] == 0x12345678u) dst[base] = acc;
}

template <int MODE, int NT>
float run(int grid, u32x4* d, u32x4* s, long long per_wg_bytes, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(512), 0, 0, d, s, per_wg_bytes / 16, reps);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(512), 0, 0, d, s, per_wg_bytes / 16, reps);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1e3f;
}

int main() {
  const long long per_wg = 640 << 10;     // bytes per workgroup and repetition (one fused-epilogue tile's stores)
  const int reps = 16;
  u32x4 *d, *s;
  hipMalloc(&d, 256 * per_wg); hipMalloc(&s, 256 * per_wg);
  hipMemset(s, 1, 256 * per_wg);
  for (int grid : {256, 64, 16, 8}) {
    const float st = run<0, 0>(grid, d, s, per_wg, reps), stn = run<0, 1>(grid, d, s, per_wg, reps), ld = run<1, 0>(grid, d, s, per_wg, reps),
                cp = run<2, 0>(grid, d, s, per_wg, reps);
    const double gb = (double)per_wg * reps / 1e9;
    printf("grid %3d: store %7.1f us (%5.1f GB/s per CU, %5.2f TB/s chip) | nt store %5.1f GB/s per CU | load %5.1f GB/s per CU | copy %5.1f GB/s per CU each way\n",
           grid, st, gb / (st * 1e-6), gb * grid / (st * 1e-6) / 1e3, gb / (stn * 1e-6), gb / (ld * 1e-6), gb / (cp * 1e-6));
  }
  return 0;
}
