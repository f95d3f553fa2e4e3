// MFMA-only ceiling under the power cap: every wave keeps two k-steps of A/B fragments (8 + 4 per step, like gemm256's 128x64 wave
// tile) in registers and issues v_mfma_f32_16x16x32_bf16 back to back, 8 waves per CU, one workgroup per CU -- no LDS, no global
// traffic inside the loop. Operands: N(0,1) bf16, or zeros. Build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

__global__ __launch_bounds__(512) void mfma_loop(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x;
  bf16x8 a[2][8], b[2][4];
  const bf16x8* p = src + (size_t)(blockIdx.x * 512 + lane) * 24;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[s][i] = p[s * 12 + i];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[s][j] = p[s * 12 + 8 + j];
  }
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 512 + lane] = s;
}

static unsigned short f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 20000;
  const size_t n = (size_t)blocks * 512 * 24 * 8;
  std::vector<unsigned short> h(n);
  bf16x8* d;
  float* o;
  hipMalloc(&d, n * 2);
  hipMalloc(&o, (size_t)blocks * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    srand(1);
    for (size_t i = 0; i < n; ++i) {
      float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
      float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
      h[i] = mode == 0 ? f2bf(g) : mode == 1 ? f2bf(g * 0.02f) : 0;     // N(0,1), N(0,0.02^2) (weight-like), zeros
    }
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 0, 0, d, o, iters / 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 0, 0, d, o, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 8 * iters * 64 * 2.0 * 16 * 16 * 32;
    printf("%s: %d blocks x 8 waves, %d iters: %.2f ms  %.1f TFLOP/s\n", mode == 0 ? "N(0,1)  " : mode == 1 ? "N(0,.02)" : "zeros   ", blocks,
           iters, ms, flops / ms / 1e9);
  }
  return 0;
}
