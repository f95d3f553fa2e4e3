// Issue cost of an LDS-DMA load for the ONLY wave of a SIMD, without compiler scheduling noise: the loop body is one inline-asm block
// on physical registers: 16 independent v_mfma_f32_16x16x32_bf16 (a[0:63]) + NL loads of 1 KiB to LDS per iteration.
// FORM 0: global_load_lds_dwordx4 v, s[base]   (saddr form, 32-bit VGPR offset)     FORM 1: buffer_load_dwordx4 v, s[rsrc], s_off offen lds
// M0UP 1: m0 is rewritten before every load (as a real kernel must, each load targets another LDS slot); 0: m0 set once.
// One workgroup of 4 waves per CU; zero operands (no power throttling). Ideal: 16 MFMAs x 16 cycles = 256 cycles per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NL, int FORM, int M0UP, int NREAD>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, float* __restrict__ out, int iters, int row_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned voff = (unsigned)((blockIdx.x * 256 + wave * 64 + (lane >> 3)) * row_bytes + (lane & 7) * 16);
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(smem) + wave * 16384);   // LDS address of this wave's 16 KiB
  const unsigned rd_addr = lds_base + lane * 16;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  float r = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile(
      "s_mov_b32 s20, %[lds]\n s_mov_b32 s21, 0\n s_mov_b32 s22, %[iters]\n s_mov_b32 m0, s20\n"
      "v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n"
      "v_mov_b32 v8, %[voff]\n v_mov_b32 v9, %[rd]\n v_mov_b32 v10, %[plo]\n v_mov_b32 v11, %[phi]\n"
      "s_nop 4\n"
      "1:\n"
      ".set i, 0\n"
      ".rept 16\n"
      "  v_mfma_f32_16x16x32_bf16 a[i*4:i*4+3], v[0:3], v[4:7], a[i*4:i*4+3]\n"
      "  .if (%c[nl] >= 1) && (i == 3)\n"
      "    .if %c[m0up]\n s_add_u32 m0, s20, 1024\n s_nop 0\n .endif\n"
      "    .if %c[form] == 0\n global_load_lds_dwordx4 v8, %[gbase] offset:0\n .elseif %c[form] == 2\n global_load_lds_dwordx4 v[10:11], off offset:0\n .else\n buffer_load_dwordx4 v8, %[rsrc], s21 offen lds\n .endif\n"
      "  .endif\n"
      "  .if (%c[nl] >= 2) && (i == 11)\n"
      "    .if %c[m0up]\n s_add_u32 m0, s20, 2048\n s_nop 0\n .endif\n"
      "    .if %c[form] == 0\n global_load_lds_dwordx4 v8, %[gbase] offset:128\n .elseif %c[form] == 2\n global_load_lds_dwordx4 v[10:11], off offset:128\n .else\n buffer_load_dwordx4 v8, %[rsrc], s21 offen offset:128 lds\n .endif\n"
      "  .endif\n"
      "  .if (%c[nl] >= 4) && ((i == 7) || (i == 15))\n"
      "    .if %c[m0up]\n s_add_u32 m0, s20, 3072\n s_nop 0\n .endif\n"
      "    .if %c[form] == 0\n global_load_lds_dwordx4 v8, %[gbase] offset:256\n .elseif %c[form] == 2\n global_load_lds_dwordx4 v[10:11], off offset:256\n .else\n buffer_load_dwordx4 v8, %[rsrc], s21 offen offset:256 lds\n .endif\n"
      "  .endif\n"
      "  .if (%c[nread] >= 1) && ((i %% %c[rdiv]) == 1)\n"
      "    ds_read_b128 v[12:15], v9 offset:(i*512)\n"
      "  .endif\n"
      "  .set i, i+1\n"
      ".endr\n"
      "s_add_u32 s21, s21, 512\n s_and_b32 s21, s21, 0x3fff\n"
      "s_waitcnt vmcnt(24)\n"
      "s_sub_u32 s22, s22, 1\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 1b\n"
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7\n s_nop 7\n"
      "v_accvgpr_read_b32 %[r], a0\n"
      : [r] "=v"(r)
      : [lds] "s"(lds_base), [iters] "s"(iters), [voff] "v"(voff), [rd] "v"(rd_addr), [gbase] "s"(src), [rsrc] "s"(rs), [nl] "n"(NL), [form] "n"(FORM),
        [plo] "v"((unsigned)((size_t)(src + voff))), [phi] "v"((unsigned)(((size_t)(src + voff)) >> 32)), [m0up] "n"(M0UP), [nread] "n"(NREAD), [rdiv] "n"(NREAD > 0 ? 16 / NREAD : 99)
      : "memory", "s20", "s21", "s22", "m0", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "a0", "a1", "a2",
        "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23",
        "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43",
        "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63",
        "scc");
#endif
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int NL, int FORM, int M0UP, int NREAD>
void run(const char* src, float* out, int iters, int row_bytes, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<NL, FORM, M0UP, NREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL((k<NL, FORM, M0UP, NREAD>), dim3(256), dim3(256), 131072, 0, src, out, iters, row_bytes);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NL, FORM, M0UP, NREAD>), dim3(256), dim3(256), 131072, 0, src, out, iters, row_bytes);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s %8.2f ms  %6.1f cycles/iteration at 2.4 GHz (ideal 256)\n", name, ms, ms * 1e-3 / iters * 2.4e9);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400000, row_bytes = 16384;
  char* src; float* out;
  const size_t bytes = (size_t)256 * 256 * row_bytes + (1 << 20);
  hipMalloc(&src, bytes); hipMemset(src, 0, bytes);
  hipMalloc(&out, 256 * 256 * 4);
  run<0, 0, 0, 0>(src, out, iters, row_bytes, "16 MFMA");
  run<0, 0, 0, 0>(src, out, iters, row_bytes, "16 MFMA (again)");
  run<0, 0, 0, 4>(src, out, iters, row_bytes, "16 MFMA + 4 ds_read_b128");
  run<1, 0, 0, 0>(src, out, iters, row_bytes, "16 MFMA + 1 global_load_lds, m0 fixed");
  run<1, 0, 1, 0>(src, out, iters, row_bytes, "16 MFMA + 1 global_load_lds, m0 rewritten");
  run<1, 1, 1, 0>(src, out, iters, row_bytes, "16 MFMA + 1 buffer_load lds,  m0 rewritten");
  run<2, 0, 1, 0>(src, out, iters, row_bytes, "16 MFMA + 2 global_load_lds, m0 rewritten");
  run<2, 1, 1, 0>(src, out, iters, row_bytes, "16 MFMA + 2 buffer_load lds,  m0 rewritten");
  run<2, 1, 1, 4>(src, out, iters, row_bytes, "16 MFMA + 2 buffer_load lds + 4 ds_read_b128  (= 128x128 wave tile mix)");
  run<2, 0, 1, 4>(src, out, iters, row_bytes, "16 MFMA + 2 global_load_lds + 4 ds_read_b128");
  run<1, 2, 1, 0>(src, out, iters, row_bytes, "16 MFMA + 1 global_load_lds (64-bit VGPR address), m0 rewritten");
  run<2, 2, 1, 0>(src, out, iters, row_bytes, "16 MFMA + 2 global_load_lds (64-bit VGPR address), m0 rewritten");
  run<4, 2, 1, 4>(src, out, iters, row_bytes, "16 MFMA + 4 global_load_lds (64-bit VGPR address) + 4 ds_read_b128");
  run<4, 0, 1, 4>(src, out, iters, row_bytes, "16 MFMA + 4 global_load_lds (SGPR base) + 4 ds_read_b128");
  run<4, 1, 1, 4>(src, out, iters, row_bytes, "16 MFMA + 4 buffer_load lds + 4 ds_read_b128");
  return 0;
}
