// What does issuing an LDS-DMA load cost the ONLY wave of a SIMD? 4 waves per CU (one per SIMD, 256 accumulator registers each like a
// 128x128 wave tile), each iteration = 128 MFMAs (one 64-k tile) with NL loads of 1 KiB interleaved one per 128/NL MFMAs, in the
// global_load_lds form (64-bit VGPR addresses) or the buffer_load ... lds form (32-bit VGPR offset + SGPR offset), plus optional
// ds_read_b128 traffic. Zero operands (no power throttling): the MFMA rate shows issue bubbles directly.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o dma_issue dma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

template <int NL, int FORM, int NR>   // NL loads and NR ds_read_b128 per 128 MFMAs; FORM 0 = global_load_lds, 1 = buffer_load lds
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, float* __restrict__ out, int iters, int row_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = bf16x8{}; b[i] = bf16x8{}; }
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  // lane -> (row, 16-B chunk) of a 128-B-wide k-slice, like a GEMM operand tile: 8 lanes per row
  const int voff = ((blockIdx.x * 256 + wave * 64 + (lane >> 3)) * row_bytes + (lane & 7) * 16);
  const char* gptr = src + voff;
  int soff = 0;
  char* dst = smem + wave * 16384;
  const char* rd = smem + wave * 16384 + lane * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int idx = g * 16 + m, i = (idx >> 3) & 7, j = idx & 7;
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (NL > 0 && (idx % (128 / (NL > 0 ? NL : 1))) == 0) {
          const int n = idx / (128 / (NL > 0 ? NL : 1));
          if (FORM == 0) __builtin_amdgcn_global_load_lds((const GLB_AS void*)(gptr + soff), (LDS_AS void*)(dst + (n & 15) * 1024), 16, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(dst + (n & 15) * 1024), 16, voff, soff, 0, 0);
        }
        if (NR > 0 && (idx % (128 / (NR > 0 ? NR : 1))) == 0) {
          const int n = idx / (128 / (NR > 0 ? NR : 1));
          const bf16x8 v = *(const bf16x8*)(rd + (n & 15) * 1024);
          if (n & 1) a[n & 7] = v; else b[n & 7] = v;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    soff = (soff + 128) & (row_bytes - 1);
    if (NL > 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NL, int FORM, int NR>
void run(const char* src, float* out, int blocks, int iters, int row_bytes, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<NL, FORM, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((k<NL, FORM, NR>), dim3(blocks), dim3(256), 65536, 0, src, out, iters / 10, row_bytes);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NL, FORM, NR>), dim3(blocks), dim3(256), 65536, 0, src, out, iters, row_bytes);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 128 * 2.0 * 16 * 16 * 32;
  printf("%-58s %8.2f ms  %7.1f TFLOP/s  (%.0f cycles per 128 MFMAs at 2.4 GHz)\n", name, ms, flops / ms / 1e9, ms * 1e-3 / iters * 2.4e9);
}

int main(int argc, char** argv) {
  const int blocks = 256, iters = argc > 1 ? atoi(argv[1]) : 4000, K = 8192, row_bytes = K * 2;
  char* src; float* out;
  const size_t bytes = (size_t)blocks * 256 * row_bytes + (size_t)iters * 128 + (1 << 20);
  hipMalloc(&src, bytes); hipMemset(src, 0, bytes);
  hipMalloc(&out, blocks * 256 * 4);
  const int it = iters;                                             // the K walk wraps inside a row
  run<0, 0, 0>(src, out, blocks, it, row_bytes, "MFMA only");
  run<0, 0, 32>(src, out, blocks, it, row_bytes, "+ 32 ds_read_b128");
  run<16, 0, 0>(src, out, blocks, it, row_bytes, "+ 16 global_load_lds");
  run<16, 1, 0>(src, out, blocks, it, row_bytes, "+ 16 buffer_load lds");
  run<16, 0, 32>(src, out, blocks, it, row_bytes, "+ 16 global_load_lds + 32 ds_read_b128");
  run<16, 1, 32>(src, out, blocks, it, row_bytes, "+ 16 buffer_load lds + 32 ds_read_b128");
  run<32, 1, 32>(src, out, blocks, it, row_bytes, "+ 32 buffer_load lds + 32 ds_read_b128");
  run<8, 1, 32>(src, out, blocks, it, row_bytes, "+  8 buffer_load lds + 32 ds_read_b128");
  run<0, 0, 0>(src, out, blocks, it, row_bytes, "MFMA only (again, clocks warm)");
  run<0, 0, 32>(src, out, blocks, it, row_bytes, "+ 32 ds_read_b128 (again)");
  run<16, 1, 0>(src, out, blocks, it, row_bytes, "+ 16 buffer_load lds (again)");
  run<16, 1, 32>(src, out, blocks, it, row_bytes, "+ 16 buffer_load lds + 32 ds_read_b128 (again)");
  return 0;
}
