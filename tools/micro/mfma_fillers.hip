// Does VALU work placed between v_mfma_f32_32x32x16_bf16 instructions hide under them? (round 4, for the attention tile body)
//   acc in AGPRs or VGPRs; four independent accumulators or one dependent chain; 0 / 4 / 8 fillers (v_fma_f32 / v_exp_f32) per MFMA gap.
// One or two waves per SIMD (blocks of 256 / 512 threads, one block per CU). Prints shader cycles per MFMA (s_memtime).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_fillers mfma_fillers.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define MF_A(i) "v_mfma_f32_32x32x16_bf16 a[" #i ":" #i "+15], v[8:11], v[12:15], a[" #i ":" #i "+15]\n"
#define MF_V(i) "v_mfma_f32_32x32x16_bf16 v[" #i ":" #i "+15], v[8:11], v[12:15], v[" #i ":" #i "+15]\n"
#define MF_QK(i) "v_mfma_f32_32x32x16_bf16 v[" #i ":" #i "+15], a[64:67], a[96:99], v[" #i ":" #i "+15]\n"      /* the attention Q K^T form */
#define MF_PV(i) "v_mfma_f32_32x32x16_bf16 a[" #i ":" #i "+15], a[64:67], v[12:15], a[" #i ":" #i "+15]\n"      /* the attention P V form */
#define F4 "v_fma_f32 v16, v16, v17, v18\n v_fma_f32 v19, v19, v17, v18\n v_fma_f32 v20, v20, v17, v18\n v_fma_f32 v21, v21, v17, v18\n"
#define E4 "v_exp_f32 v16, v16\n v_exp_f32 v19, v19\n v_exp_f32 v20, v20\n v_exp_f32 v21, v21\n"

#define KERNEL(NAME, BODY, CLOB)                                                                           \
  __global__ void NAME(long long* out, int iters) {                                                        \
    long long t0 = __builtin_readcyclecounter();                                                           \
    for (int it = 0; it < iters; ++it) asm volatile(BODY ::: CLOB);                                        \
    long long t1 = __builtin_readcyclecounter();                                                           \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[threadIdx.x >> 6] = t0; out[8 + (threadIdx.x >> 6)] = t1; }           \
  }
#define CLOB_A "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21"
#define CLOB_V "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95"

// 8 MFMAs per body
KERNEL(a_ind_0, REP8(MF_A(0) MF_A(16) MF_A(32) MF_A(48)), CLOB_A)                    // 32 MFMAs per body (4 x 8)
KERNEL(a_ind_f4, REP8(MF_A(0) F4 MF_A(16) F4 MF_A(32) F4 MF_A(48) F4), CLOB_A)
KERNEL(a_ind_f8, REP8(MF_A(0) F4 F4 MF_A(16) F4 F4 MF_A(32) F4 F4 MF_A(48) F4 F4), CLOB_A)
KERNEL(a_ind_e4, REP8(MF_A(0) E4 MF_A(16) E4 MF_A(32) E4 MF_A(48) E4), CLOB_A)
KERNEL(a_dep_0, REP8(MF_A(0) MF_A(0) MF_A(0) MF_A(0)), CLOB_A)
KERNEL(a_dep_f8, REP8(MF_A(0) F4 F4 MF_A(0) F4 F4 MF_A(0) F4 F4 MF_A(0) F4 F4), CLOB_A)
KERNEL(v_ind_0, REP8(MF_V(32) MF_V(48) MF_V(64) MF_V(80)), CLOB_V)
KERNEL(v_ind_f4, REP8(MF_V(32) F4 MF_V(48) F4 MF_V(64) F4 MF_V(80) F4), CLOB_V)
KERNEL(v_ind_f8, REP8(MF_V(32) F4 F4 MF_V(48) F4 F4 MF_V(64) F4 F4 MF_V(80) F4 F4), CLOB_V)
KERNEL(v_dep_0, REP8(MF_V(32) MF_V(32) MF_V(32) MF_V(32)), CLOB_V)
KERNEL(v_dep_f8, REP8(MF_V(32) F4 F4 MF_V(32) F4 F4 MF_V(32) F4 F4 MF_V(32) F4 F4), CLOB_V)
#define CLOB_AV CLOB_V, "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a96","a97","a98","a99"
KERNEL(qk_form, REP8(MF_QK(32) MF_QK(32) MF_QK(48) MF_QK(48)), CLOB_AV)
KERNEL(qk_form_f4, REP8(MF_QK(32) F4 MF_QK(32) F4 MF_QK(48) F4 MF_QK(48) F4), CLOB_AV)
KERNEL(pv_form, REP8(MF_PV(0) MF_PV(16) MF_PV(32) MF_PV(48)), CLOB_AV)
KERNEL(pv_form_f4, REP8(MF_PV(0) F4 MF_PV(16) F4 MF_PV(32) F4 MF_PV(48) F4), CLOB_AV)
KERNEL(fill_only_f8, REP8(F4 F4 F4 F4 F4 F4 F4 F4), CLOB_A)

int main() {
  long long* d;
  hipMalloc(&d, 16 * 8);
  struct { const char* name; void (*k)(long long*, int); } ks[] = {
      {"AGPR acc, 4 independent, no fillers", a_ind_0}, {"AGPR acc, 4 independent, 4 v_fma per gap", a_ind_f4},
      {"AGPR acc, 4 independent, 8 v_fma per gap", a_ind_f8}, {"AGPR acc, 4 independent, 4 v_exp per gap", a_ind_e4},
      {"AGPR acc, dependent chain, no fillers", a_dep_0}, {"AGPR acc, dependent chain, 8 v_fma per gap", a_dep_f8},
      {"VGPR acc, 4 independent, no fillers", v_ind_0}, {"VGPR acc, 4 independent, 4 v_fma per gap", v_ind_f4},
      {"VGPR acc, 4 independent, 8 v_fma per gap", v_ind_f8}, {"VGPR acc, dependent chain, no fillers", v_dep_0},
      {"VGPR acc, dependent chain, 8 v_fma per gap", v_dep_f8}, {"Q K^T form (A, B in AGPRs, D in VGPRs), chains of 2", qk_form}, {"Q K^T form, 4 v_fma per gap", qk_form_f4},
      {"P V form (A AGPR, B VGPR, D AGPR)", pv_form}, {"P V form, 4 v_fma per gap", pv_form_f4},
      {"(64 v_fma alone, per 8 'gaps')", fill_only_f8}};
  const int iters = 200;
  for (int threads : {256, 512}) {
    printf("---- %d waves per SIMD (blocks of %d threads, one per CU)\n", threads / 256, threads);
    for (auto& e : ks) {
      long long h[16];
      hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, d, 2);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 0, 0, d, iters);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      // wave w of the block runs on SIMD w % 4: waves 0 and 4 share a SIMD. Per wave: its own span; per SIMD: first start to last end
      const int nw = threads / 64;
      long long lo = h[0], hi = h[8];
      for (int w = 0; w < nw; w += 4) { lo = h[w] < lo ? h[w] : lo; hi = h[8 + w] > hi ? h[8 + w] : hi; }
      printf("%-56s %7.1f cycles per MFMA in wave 0, %7.1f per MFMA of the SIMD (%d waves)\n", e.name, (double)(h[8] - h[0]) / (iters * 32.0),
             (double)(hi - lo) / (iters * 32.0 * (nw / 4)), nw / 4);
    }
  }
  return 0;
}
