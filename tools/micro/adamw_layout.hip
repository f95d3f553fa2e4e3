// AdamW memory layouts under the same arithmetic: SoA (five streams: p, g, m, v fp32 + bf16 copy) vs AoS state (one stream of
// {p4, m4, v4} 48-byte records + g + bf16 copy = three streams). Chunk-per-workgroup walk in both. hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

__device__ inline unsigned pk(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }
__device__ inline void upd(f4& p, const f4 g, f4& m, f4& v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float mm = 0.9f * m[r] + 0.1f * g[r], vv = 0.999f * v[r] + 0.001f * g[r] * g[r];
    p[r] -= 1e-4f * mm / (sqrtf(vv) + 1e-8f);
    m[r] = mm; v[r] = vv;
  }
}
__global__ __launch_bounds__(256) void soa(f4* p, const f4* g, f4* m, f4* v, u2* p16, long long n4) {
  const long long chunk = (((n4 + gridDim.x - 1) / gridDim.x) + 255) & ~255LL, lo = blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    f4 pp = p[i], mm = m[i], vv = v[i];
    upd(pp, __builtin_nontemporal_load(g + i), mm, vv);
    p[i] = pp; m[i] = mm; v[i] = vv;
    p16[i] = u2{pk(pp[0], pp[1]), pk(pp[2], pp[3])};
  }
}
__global__ __launch_bounds__(256) void aos(f4* st, const f4* g, u2* p16, long long n4) {
  const long long chunk = (((n4 + gridDim.x - 1) / gridDim.x) + 255) & ~255LL, lo = blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    f4 pp = st[3 * i], mm = st[3 * i + 1], vv = st[3 * i + 2];
    upd(pp, __builtin_nontemporal_load(g + i), mm, vv);
    st[3 * i] = pp; st[3 * i + 1] = mm; st[3 * i + 2] = vv;
    p16[i] = u2{pk(pp[0], pp[1]), pk(pp[2], pp[3])};
  }
}
// AoS at 1 KiB granularity: per 64 consecutive float4 groups the record is [p x64 | m x64 | v x64] -> every access stays a 1 KiB run
__global__ __launch_bounds__(256) void aos_blk(f4* st, const f4* g, u2* p16, long long n4) {
  const long long chunk = (((n4 + gridDim.x - 1) / gridDim.x) + 255) & ~255LL, lo = blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const long long b = (i >> 6) * 192 + (i & 63);
    f4 pp = st[b], mm = st[b + 64], vv = st[b + 128];
    upd(pp, __builtin_nontemporal_load(g + i), mm, vv);
    st[b] = pp; st[b + 64] = mm; st[b + 128] = vv;
    p16[i] = u2{pk(pp[0], pp[1]), pk(pp[2], pp[3])};
  }
}
int main() {
  const long long n = 202383360, n4 = n / 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int trial = 0; trial < 3; ++trial) {
    float *p, *g, *m, *v, *st; void* p16;
    hipMalloc(&p, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4); hipMalloc(&p16, n * 2); hipMalloc(&st, n * 12);
    hipMemset(p, 0, n * 4); hipMemset(g, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4); hipMemset(st, 0, n * 12);
    float ms[3];
    for (int k = 0; k < 3; ++k) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) {
          if (k == 0) hipLaunchKernelGGL(soa, dim3(8192), dim3(256), 0, 0, (f4*)p, (const f4*)g, (f4*)m, (f4*)v, (u2*)p16, n4);
          if (k == 1) hipLaunchKernelGGL(aos, dim3(8192), dim3(256), 0, 0, (f4*)st, (const f4*)g, (u2*)p16, n4);
          if (k == 2) hipLaunchKernelGGL(aos_blk, dim3(8192), dim3(256), 0, 0, (f4*)st, (const f4*)g, (u2*)p16, n4);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[k], e0, e1);
      }
    }
    printf("trial %d: SoA %.3f ms %.2f TB/s | AoS(48 B records) %.3f ms %.2f TB/s | AoS(1 KiB blocks) %.3f ms %.2f TB/s\n", trial, ms[0] / 10,
           n * 30.0 / (ms[0] / 10) / 1e9, ms[1] / 10, n * 30.0 / (ms[1] / 10) / 1e9, ms[2] / 10, n * 30.0 / (ms[2] / 10) / 1e9);
    // leak the buffers of this trial on purpose: the next trial gets a different placement
  }
  return 0;
}
