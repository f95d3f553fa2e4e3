// LDS read throughput per CU for the three fragment-read forms: ds_read_b128, ds_read_b64 and the transposing ds_read_b64_tr_b16.
// One workgroup of W waves per CU, every wave issues UNROLL independent reads per loop trip from conflict-free addresses
// (lane-linear), nothing else in the loop. Prints bytes / cycle / CU. Build: hipcc --offload-arch=gfx950 -O3 lds_read_rate.hip -o lds_read_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define LDS_AS __attribute__((address_space(3)))

template <int FORM>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((unsigned*)smem)[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned acc = 0;
  const char* base = smem + (wave & 3) * 16384 + lane * (FORM == 0 ? 16 : 8);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const char* a = base + u * (FORM == 0 ? 1024 : 512) + ((it & 1) << 13);
      // inline assembly: the compiler must neither hoist the (loop-invariant) reads nor merge them
      const unsigned la = (unsigned)(unsigned long long)(LDS_AS const char*)a;
      if (FORM == 0) { u32x4 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(7)" : "=v"(v) : "v"(la)); acc += v[0] ^ v[3]; }
      else if (FORM == 1) { u32x2 v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(7)" : "=v"(v) : "v"(la)); acc += v[0] ^ v[1]; }
      else { u32x2 v; asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(7)" : "=v"(v) : "v"(la)); acc += v[0] ^ v[1]; }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int FORM>
void run(const char* name, int waves, unsigned* out, long long* cyc) {
  const int iters = 20000;
  hipFuncSetAttribute((const void*)k<FORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k<FORM>, dim3(256), dim3(64 * waves), 65536, 0, out, iters, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<FORM>, dim3(256), dim3(64 * waves), 65536, 0, out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double bytes = (double)iters * 8 * (FORM == 0 ? 1024 : 512) * waves;
  printf("%-22s %2d waves/CU: %7.1f GB/s/CU (wall clock) = %5.1f B/clk at 2.4 GHz\n", name, waves, bytes / (ms * 1e6), bytes / (ms * 1e6) / 2.4);
  (void)c;
}

int main() {
  unsigned* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  for (int w : {4, 8, 16}) {
    run<0>("ds_read_b128", w, out, cyc);
    run<1>("ds_read_b64", w, out, cyc);
    run<2>("ds_read_b64_tr_b16", w, out, cyc);
  }
  return 0;
}
