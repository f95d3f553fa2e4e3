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

// BAR = number of s_barrier per 64 MFMAs (0, 2, 4, 8): the issue cost of gemm256's phase barriers without any memory traffic
template <int BAR>
__global__ __launch_bounds__(512) void mfma_bar_loop(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
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
  if (BAR && (threadIdx.x >> 8)) __builtin_amdgcn_s_barrier();       // stagger the two wave rows like gemm256
  for (int it = 0; it < iters; ++it) {
    // four quadrants (4 x 2 blocks) x 2 k-steps = 64 MFMAs, like one K-tile of gemm256
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[(q >> 1) * 4 + i][(q & 1) * 2 + j] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][(q >> 1) * 4 + i], b[s][(q & 1) * 2 + j], acc[(q >> 1) * 4 + i][(q & 1) * 2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (BAR >= 4 || (BAR == 2 && (q & 1))) __builtin_amdgcn_s_barrier();
      if (BAR >= 8) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (BAR && !(threadIdx.x >> 8)) __builtin_amdgcn_s_barrier();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 512 + lane] = s;
}

// Same 64 MFMAs per iteration (8 x 4 output blocks, 2 k-steps) issued quadrant by quadrant: a quadrant is QI x QJ blocks, inside it
// the order is k-step -> row block -> column block (KIN = 0) or row -> column -> k-step (KIN = 1). Which order costs the least power?
template <int QI, int QJ, int KIN>
__global__ __launch_bounds__(512) void mfma_order_loop(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
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
    for (int qi = 0; qi < 8 / QI; ++qi)
#pragma unroll
      for (int qj = 0; qj < 4 / QJ; ++qj) {
        if (KIN == 0) {
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < QI; ++i)
#pragma unroll
              for (int j = 0; j < QJ; ++j)
                acc[qi * QI + i][qj * QJ + j] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][qi * QI + i], b[s][qj * QJ + j], acc[qi * QI + i][qj * QJ + j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < QI; ++i)
#pragma unroll
            for (int j = 0; j < QJ; ++j)
#pragma unroll
              for (int s = 0; s < 2; ++s)
                acc[qi * QI + i][qj * QJ + j] =
                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][qi * QI + i], b[s][qj * QJ + j], acc[qi * QI + i][qj * QJ + j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 512 + lane] = s;
}

// the same 128 x 64 wave tile with v_mfma_f32_32x32x16_bf16: 4 x 2 blocks of 32 x 32 (16 accumulator registers each), 4 k-steps of 16 per
// 64-k tile = 32 MFMAs of 32768 flops per iteration (the same flops as the 64 MFMAs of the 16x16x32 loops)
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
__global__ __launch_bounds__(512) void mfma32_loop(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x;
  bf16x8 a[4][4], b[4][2];
  const bf16x8* p = src + (size_t)(blockIdx.x * 512 + lane) * 24;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a[s][i] = p[s * 6 + i];
#pragma unroll
    for (int j = 0; j < 2; ++j) b[s][j] = p[s * 6 + 4 + j];
  }
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  out[blockIdx.x * 512 + lane] = sum;
}

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
    if (mode != 1) {
      auto run = [&](auto kern, const char* name) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, o, iters / 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, o, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        printf("    gemm256-shaped MFMA stream, %s: %.1f TFLOP/s\n", name, flops / t / 1e9);
      };
      run(mfma32_loop, "32x32x16 MFMA, 4x2 blocks");
      run(mfma_order_loop<4, 2, 0>, "order 4x2 k-outer    ");
      run(mfma32_loop, "32x32x16 MFMA (again)   ");
      run(mfma_order_loop<8, 4, 0>, "order 8x4 k-outer    ");
      run(mfma_order_loop<4, 4, 0>, "order 4x4 k-outer    ");
      run(mfma_order_loop<4, 2, 0>, "order 4x2 k-outer    ");
      run(mfma_order_loop<2, 2, 0>, "order 2x2 k-outer    ");
      run(mfma_order_loop<4, 1, 0>, "order 4x1 k-outer    ");
      run(mfma_order_loop<2, 1, 0>, "order 2x1 k-outer    ");
      run(mfma_order_loop<1, 4, 0>, "order 1x4 k-outer    ");
      run(mfma_order_loop<1, 2, 0>, "order 1x2 k-outer    ");
      run(mfma_order_loop<8, 4, 1>, "order 8x4 k-inner    ");
      run(mfma_order_loop<4, 2, 1>, "order 4x2 k-inner    ");
      run(mfma_bar_loop<0>, "no barrier           ");
      run(mfma_bar_loop<2>, "2 barriers / 64 MFMA ");
      run(mfma_bar_loop<4>, "4 barriers / 64 MFMA ");
      run(mfma_bar_loop<8>, "8 barriers / 64 MFMA ");
    }
  }
  return 0;
}
