// mfma_rate.hip -- what does one SIMD sustain on v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 with 1, 2 or 4 independent accumulator chains per wave
// and 1 or 2 waves per SIMD?  (The decode's team kernels contract 16-row tiles with TWO chains per wave and two waves per SIMD and ran at half the
// rate the instruction's pass count suggests.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_rate.hip -o tools/micro/kp_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

template <int NA>
__global__ void k16(float* out, long long* cyc, int iters) {
  f32x4 acc[NA];
  for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NA>
__global__ void k32(float* out, long long* cyc, int iters) {
  f32x16 acc[NA];
  for (int i = 0; i < NA; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
template <typename K> static void run(const char* name, K kern, int na, int threads) {
  float* d_o; long long* d_c; CK(hipMalloc(&d_o, 1024 * 4)); CK(hipMalloc(&d_c, 64 * 8));
  const int iters = 2000;
  hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d_o, d_c, iters); CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d_o, d_c, iters); CK(hipDeviceSynchronize());
  std::vector<long long> c(threads / 64); CK(hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost));
  long long mx = 0; for (auto v : c) mx = v > mx ? v : mx;
  const double per_simd = (double)(threads / 64 > 4 ? 2 : 1) * iters * 8 * na;       // MFMAs one SIMD executed (waves w and w + 4 share a SIMD)
  printf("%s, %d chain(s) per wave, %d wave(s) per SIMD: %.1f shader cycles per MFMA per SIMD\n", name, na, threads / 64 > 4 ? 2 : 1, (double)mx / per_simd);
  CK(hipFree(d_o)); CK(hipFree(d_c));
}
int main() {
  run("v_mfma_f32_16x16x4_f32", k16<1>, 1, 256); run("v_mfma_f32_16x16x4_f32", k16<2>, 2, 256); run("v_mfma_f32_16x16x4_f32", k16<4>, 4, 256);
  run("v_mfma_f32_16x16x4_f32", k16<1>, 1, 512); run("v_mfma_f32_16x16x4_f32", k16<2>, 2, 512); run("v_mfma_f32_16x16x4_f32", k16<4>, 4, 512);
  run("v_mfma_f32_32x32x2_f32", k32<1>, 1, 256); run("v_mfma_f32_32x32x2_f32", k32<2>, 2, 256);
  run("v_mfma_f32_32x32x2_f32", k32<1>, 1, 512); run("v_mfma_f32_32x32x2_f32", k32<2>, 2, 512);
  return 0;
}
