// mla_calib_mfma: what the matrix cores of THIS box sustain on random bf16 operands under ITS power cap -- the yardstick bench.py
// prints next to the GEMM roofline fraction (round 6). The step's GEMMs run at the board power limit, so the same
// build measures 567-598 ms per step across the pool's boxes (profiles/r5_box_spread.txt) and `roofline.frac` (against the nominal
// 2.5 PFLOP/s) moves with the box; `frac_of_box_ceiling` = achieved / this stream's rate does not.
// The stream: every wave holds two k-steps of A / B fragments (8 + 4 per step -- the 128 x 64 wave tile of gemm256) in registers and
// issues v_mfma_f32_16x16x32_bf16 back to back, 8 waves per workgroup, no LDS, no memory traffic inside the loop (the kernel of
// tools/micro/mfma_power.hip behind one export). Work per launch = blocks x 8 waves x iters x 64 MFMAs x 16 384 flop.
#include "common.h"

namespace {

__global__ __launch_bounds__(512) void calib_mfma_kernel(const bf16x8_t* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x;
  bf16x8_t a[2][8], b[2][4];
  const bf16x8_t* p = src + (size_t)lane * 24;       // the same 512 x 24 fragments for every workgroup: 192 KiB of random operands
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[s][i] = p[s * 12 + i];
#pragma unroll
    for (int j = 0; j < 4; ++j) b[s][j] = p[s * 12 + 8 + j];
  }
  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {                     // four quadrants (4 x 2 fragments) x 2 k-steps = 64 MFMAs, like one K-tile of gemm256
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[(q >> 1) * 4 + i][(q & 1) * 2 + j] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][(q >> 1) * 4 + i], b[s][(q & 1) * 2 + j], acc[(q >> 1) * 4 + i][(q & 1) * 2 + j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[(size_t)blockIdx.x * 512 + lane] = s;
}

}  // namespace

// operands: 512 x 24 x 8 bf16 (196 608 bytes, 16-B aligned; the caller fills them, e.g. N(0, 1)); out: blocks x 512 floats (a sink that keeps
// the accumulators alive). Returns the flop count of one launch through *flops_per_launch (may be NULL).
extern "C" int mla_calib_mfma(const void* operands, float* out, int blocks, int iters, double* flops_per_launch, hipStream_t stream) {
  MLA_CHECK_ARG(operands && out && blocks > 0 && blocks <= 65536 && iters > 0, "mla_calib_mfma: null pointer or bad shape");
  MLA_CHECK_ARG((((uintptr_t)operands) & 15) == 0, "mla_calib_mfma: operands must be 16-B aligned");
  if (flops_per_launch) *flops_per_launch = (double)blocks * 8.0 * (double)iters * 64.0 * 16384.0;
  hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(512), 0, stream, (const bf16x8_t*)operands, out, iters);
  MLA_LAUNCH_CHECK();
}
