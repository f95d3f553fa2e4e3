// LDS read rates on gfx950 (round 4, for the attention kernels): how many cycles of the CU's LDS pipe does one wave-wide
// ds_read_b128 (1 KiB) / ds_read_b64 (512 B) / ds_read_b64_tr_b16 (512 B, the transposed operand read) cost when all four SIMDs
// of a CU stream them? Prints cycles per instruction per CU and bytes per clock.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_rates lds_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(NAME, INSTR)                                                                               \
  __global__ void NAME(long long* out, int iters) {                                                       \
    extern __shared__ char smem[];                                                                        \
    const unsigned addr = (unsigned)(unsigned long long)smem + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096; \
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((int*)smem)[i] = i;                            \
    __syncthreads();                                                                                      \
    long long t0 = __builtin_readcyclecounter();                                                          \
    for (int it = 0; it < iters; ++it) asm volatile(REP16(INSTR) "s_waitcnt lgkmcnt(0)\n" ::"v"(addr) : "v8", "v9", "v10", "v11", "memory"); \
    long long t1 = __builtin_readcyclecounter();                                                          \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[threadIdx.x >> 6] = t0; out[8 + (threadIdx.x >> 6)] = t1; } \
  }
KERNEL(k_b128, "ds_read_b128 v[8:11], %0\n")
KERNEL(k_b64, "ds_read_b64 v[8:9], %0\n")
KERNEL(k_tr, "ds_read_b64_tr_b16 v[8:9], %0\n")

int main() {
  long long* d;
  (void)hipMalloc(&d, 16 * 8);
  struct { const char* name; void (*k)(long long*, int); int bytes; } ks[] = {
      {"ds_read_b128 (lane * 16 B: conflict-free)", k_b128, 1024}, {"ds_read_b64 (lane * 16 B stride)", k_b64, 512},
      {"ds_read_b64_tr_b16 (lane * 16 B stride)", k_tr, 512}};
  const int iters = 500;
  for (int threads : {256, 512}) {
    printf("---- %d waves per CU\n", threads / 64);
    for (auto& e : ks) {
      long long h[16];
      hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 65536, 0, d, 2);
      (void)hipDeviceSynchronize();
      hipLaunchKernelGGL(e.k, dim3(256), dim3(threads), 65536, 0, d, iters);
      (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      const int nw = threads / 64;
      long long lo = h[0], hi = h[8];
      for (int w = 0; w < nw; ++w) { lo = h[w] < lo ? h[w] : lo; hi = h[8 + w] > hi ? h[8 + w] : hi; }
      const double per = (double)(hi - lo) / (iters * 16.0 * nw);
      printf("%-46s %6.2f cycles of the CU per instruction = %6.1f B/clk\n", e.name, per, e.bytes / per);
    }
  }
  return 0;
}
