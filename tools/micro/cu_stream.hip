// Micro-benchmark: how fast can ONE workgroup (512 threads on one CU) stream a 256 KB weight matrix per "layer" out of L2,
// layer after layer with a workgroup barrier in between -- the bound of a row-split fused MLP (a few rows x all columns per
// workgroup, 7 dependent k=1 layers in one launch).  Variants: all 32 float4 loads of a layer in flight at once, or the next
// layer's loads issued before the current layer's reduction (prefetch).  NWG workgroups stream the SAME weights.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LAYERS = 7, KW = 256, NW = 256;        // 256 x 256 floats per layer

template <bool PREFETCH>
__global__ void __launch_bounds__(512) k_stream(const float* __restrict__ w, float* out, int reps) {
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
    if (PREFETCH) load(cur, 0);
    for (int l = 0; l < LAYERS; ++l) {
      if (!PREFETCH) load(cur, l);
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 32; ++u) s += cur[u] * (acc + 1.0f);
      if (PREFETCH && l + 1 < LAYERS) load(nxt, l + 1);
      red[wave * 256 + lane * 4] = s[0] + s[1] + s[2] + s[3];
      __syncthreads();
      float t = 0.f;
      for (int q = 0; q < 8; ++q) t += red[q * 256 + lane * 4];
      acc = t * 1e-9f;
      __syncthreads();
      if (PREFETCH) {
#pragma unroll
        for (int u = 0; u < 32; ++u) cur[u] = nxt[u];
      }
    }
  }
  if (tid == 0) out[blockIdx.x] = acc;
}

int main() {
  float *w, *out;
  hipMalloc(&w, (size_t)LAYERS * KW * NW * 4); hipMalloc(&out, 4096);
  hipMemset(w, 0, (size_t)LAYERS * KW * NW * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 200;
  for (int nwg : {1, 8, 32}) {
    for (int pf = 0; pf < 2; ++pf) {
      for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        if (pf) hipLaunchKernelGGL(k_stream<true>, dim3(nwg), dim3(512), 0, 0, w, out, reps);
        else hipLaunchKernelGGL(k_stream<false>, dim3(nwg), dim3(512), 0, 0, w, out, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps / LAYERS;
      printf("%2d workgroups, %s: %.2f us per 256 KB layer = %.1f GB/s per CU\n", nwg, pf ? "next layer prefetched" : "loads at layer start ", us, 262144.0 / us / 1e3);
    }
  }
  return 0;
}
