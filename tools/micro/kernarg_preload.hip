// Micro-benchmark: does kernarg preloading (-mllvm -amdgpu-kernarg-preload-count=N) shorten a chain of tiny dependent kernels?
// Each kernel: 32 workgroups x 512 threads, one dependent global load through a pointer argument, one store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Big { const float* a; float* b; int n; int pad[60]; };
__global__ void __launch_bounds__(512) k_small(const float* a, float* b, int n, int off) {
  const int i = blockIdx.x * 512 + threadIdx.x;
  if (i < n) b[i] = a[i] + (float)off;
}
__global__ void __launch_bounds__(512) k_big(Big p) {
  const int i = blockIdx.x * 512 + threadIdx.x;
  if (i < p.n) p.b[i] = p.a[i] + (float)p.pad[7];
}
int main() {
  const int n = 32 * 512, iters = 2000;
  float *a, *b; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMemset(a, 0, n * 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  Big P; P.a = a; P.b = b; P.n = n; for (int i = 0; i < 60; ++i) P.pad[i] = i;
  for (int variant = 0; variant < 2; ++variant) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 100; ++i) {
      if (variant == 0) hipLaunchKernelGGL(k_small, dim3(32), dim3(512), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n, i);
      else { Big q = P; q.a = (i & 1) ? b : a; q.b = (i & 1) ? a : b; hipLaunchKernelGGL(k_big, dim3(32), dim3(512), 0, s, q); }
    }
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < iters / 100; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.2f us per dependent kernel\n", variant == 0 ? "4 scalar args (16 B)" : "260-byte struct arg", ms * 1000.f / iters);
  }
  return 0;
}
