// Micro-benchmark 2: the same one-workgroup-per-CU weight stream as cu_stream.hip, with the footprint as a parameter: NL "layers" of
// 256 KB streamed cyclically by NWG workgroups (all the SAME weights).  1.8 MB (7 layers) stays in one XCD's 4 MB L2; 11.5 MB (44
// passes = a whole decode chain piece) does not, so every pass comes from the Infinity Cache / HBM through the fabric.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KW = 256, NW = 256;

template <bool BARRIER>
__global__ void __launch_bounds__(512) k_stream(const float* __restrict__ w, float* out, int reps, int nl) {
  __shared__ float red[8 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc = 0.f;
  f32x4 cur[32], nxt[32];
  auto load = [&](f32x4 (&v)[32], int layer) {
    const float* p = w + (size_t)layer * KW * NW + (size_t)(wave * 32) * NW + lane * 4;
#pragma unroll
    for (int u = 0; u < 32; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)u * NW);
  };
  for (int r = 0; r < reps; ++r) {
    load(cur, 0);
    for (int l = 0; l < nl; ++l) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 32; ++u) s += cur[u] * (acc + 1.0f);
      if (l + 1 < nl) load(nxt, l + 1);
      if (BARRIER) {
        red[wave * 256 + lane * 4] = s[0] + s[1] + s[2] + s[3];
        __syncthreads();
        float t = 0.f;
        for (int q = 0; q < 8; ++q) t += red[q * 256 + lane * 4];
        acc = t * 1e-9f;
        __syncthreads();
      } else acc += (s[0] + s[1] + s[2] + s[3]) * 1e-9f;
#pragma unroll
      for (int u = 0; u < 32; ++u) cur[u] = nxt[u];
    }
  }
  if (tid == 0) out[blockIdx.x] = acc;
}

int main() {
  float *w, *out;
  const int maxl = 64;
  hipMalloc(&w, (size_t)maxl * KW * NW * 4); hipMalloc(&out, 4096);
  hipMemset(w, 0, (size_t)maxl * KW * NW * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nl : {7, 14, 28, 44}) for (int nwg : {1, 16, 32}) for (int bar = 0; bar < 2; ++bar) {
    const int reps = 400 / nl + 1;
    for (int it = 0; it < 2; ++it) {
      hipEventRecord(e0);
      if (bar) hipLaunchKernelGGL(k_stream<true>, dim3(nwg), dim3(512), 0, 0, w, out, reps, nl);
      else hipLaunchKernelGGL(k_stream<false>, dim3(nwg), dim3(512), 0, 0, w, out, reps, nl);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps / nl;
    printf("%2d layers (%.1f MB) %2d workgroups %s: %.2f us per 256 KB = %.1f GB/s per CU\n", nl, nl * 0.262144, nwg, bar ? "barrier   " : "no barrier", us, 262144.0 / us / 1e3);
  }
  return 0;
}
