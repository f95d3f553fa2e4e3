// Cost of a dependent kernel boundary on one stream: N launches of an (almost) empty kernel, wall time per launch; the same chain
// replayed from a captured hipGraph. Build: hipcc --offload-arch=gfx950 -O3 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void tiny(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.f; }
__global__ void busy(float* x, int iters) {   // ~20 us of work on every CU, so that the host is always ahead
  float a = x[threadIdx.x & 63];
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  if (a == 123.f) x[1] = a;
}
int main() {
  float* x; hipMalloc(&x, 4096); hipMemset(x, 0, 4096);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 2000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, s);
      for (int i = 0; i < N; ++i) {
        if (mode == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, x);
        else hipLaunchKernelGGL(busy, dim3(512), dim3(256), 0, s, x, 4000);
      }
      hipEventRecord(e1, s); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s: %.2f us per launch (stream)\n", mode ? "busy kernel (512 x 256 threads)" : "empty kernel", ms * 1e3 / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) {
      if (mode == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, x);
      else hipLaunchKernelGGL(busy, dim3(512), dim3(256), 0, s, x, 4000);
    }
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s: %.2f us per launch (hipGraph replay)\n", mode ? "busy kernel (512 x 256 threads)" : "empty kernel", ms * 1e3 / N);
    }
  }
  return 0;
}
