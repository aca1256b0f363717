// mfma_feed_lab.hip -- how much of a SIMD's fp32 matrix pipe does a wave keep busy when its MFMA operands arrive DURING the loop: B as ds_read_b128 from an LDS-resident
// weight slice (xcone_kernel's GEMM layers, round 6), A as global_load_dwordx4 through a ring of eight k-groups?  mfma_rate.hip says 32.0 cycles per
// v_mfma_f32_16x16x4_f32 with register operands in every configuration; xcone_kernel's loop measured ~40 (two waves per SIMD) / ~45 (one).  This lab adds the feeds one
// at a time.  One "k-group" = 16 channels = 8 MFMAs (two 16-column tiles x four k-steps), as in the kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_feed_lab.hip -o tools/micro/kp_mfma_feed_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s @%d\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// MODE bit 0: B from LDS (one k-group ahead), bit 1: A from global memory (ring of 8), bit 2: scheduling barriers pin the requests (the kernel's form),
// bit 3: four accumulator chains instead of two, bit 4: requests spread behind single MFMAs (sched_group_barrier)
template <int MODE>
__global__ void __launch_bounds__(512) lab(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ out, long long* __restrict__ cyc, int groups) {
  __shared__ __attribute__((aligned(16))) float wlds[2 * 48 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 48 * 256 / 4; i += blockDim.x) reinterpret_cast<f32x4*>(wlds)[i] = reinterpret_cast<const f32x4*>(W)[i];
  __syncthreads();
  const float* wl = wlds + lane * 4;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc0 = z4, acc1 = z4, acc2 = z4, acc3 = z4;
  f32x4 ring[8];
  const char* ab = reinterpret_cast<const char*>(A) + ((size_t)(blockIdx.x * 8 + wave) * 16 + (lane & 15)) * 1024 * 4 + (lane >> 4) * 16;   // 16 rows of 1024 floats per wave
#pragma unroll
  for (int i = 0; i < 8; ++i) ring[i] = (MODE & 2) ? *reinterpret_cast<const f32x4*>(ab + 64 * i) : f32x4{1.f + lane * 1e-3f, 0.5f, 0.25f, 0.125f};
  f32x4 b0 = *reinterpret_cast<const f32x4*>(wl), b1 = *reinterpret_cast<const f32x4*>(wl + 48 * 256);
  __syncthreads();
  const long long t0 = clock64();
  for (int g0 = 0; g0 < groups; g0 += 8) {
    const unsigned ao = (unsigned)(((g0 + 8) & 63) * 64);                 // the next chunk of eight k-groups of this wave's rows (wraps inside the row: L2 / L1 resident)
    const float* wg = wl + ((g0 + 1) % 40) * 256;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x4 nb0 = b0, nb1 = b1;
      if (MODE & 1) { nb0 = *reinterpret_cast<const f32x4*>(wg + i * 256); nb1 = *reinterpret_cast<const f32x4*>(wg + (48 + i) * 256); }
      if (MODE & 4) __builtin_amdgcn_sched_barrier(0);
      const f32x4 a = ring[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if ((MODE & 8) && (e & 1)) {
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0[e], acc2, 0, 0, 0);
          acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b1[e], acc3, 0, 0, 0);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0[e], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b1[e], acc1, 0, 0, 0);
        }
      }
      if (MODE & 2) ring[i] = *reinterpret_cast<const f32x4*>(ab + ao + 64 * i);
      if (MODE & 16) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
      }
      if (MODE & (4 | 16)) __builtin_amdgcn_sched_barrier(0);
      b0 = nb0; b1 = nb1;
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + tid] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + ring[0][0];
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <typename K> static void run(const char* name, K kern, int threads, int blocks) {
  const int groups = 24 * 64;
  float *dA, *dW, *dO; long long* dC;
  const size_t na = (size_t)blocks * 8 * 16 * 1024;
  CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dW, 2 * 48 * 256 * 4)); CK(hipMalloc(&dO, (size_t)blocks * 512 * 4)); CK(hipMalloc(&dC, (size_t)blocks * 8 * 8));
  CK(hipMemset(dA, 0, na * 4)); CK(hipMemset(dW, 0, 2 * 48 * 256 * 4));
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dA, dW, dO, dC, groups); CK(hipDeviceSynchronize()); }
  const int nw = threads / 64;
  std::vector<long long> c((size_t)blocks * nw); CK(hipMemcpy(c.data(), dC, c.size() * 8, hipMemcpyDeviceToHost));
  long long mx = 0; double sum = 0; for (auto v : c) { mx = v > mx ? v : mx; sum += (double)v; }
  const double per_simd = (double)(nw > 4 ? 2 : 1) * groups * 8;
  printf("%-86s %d wave(s)/SIMD, %3d workgroup(s): %.1f cycles per MFMA per SIMD (slowest wave), %.1f (mean)\n", name, nw > 4 ? 2 : 1, blocks, (double)mx / per_simd, sum / c.size() / per_simd);
  CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dO)); CK(hipFree(dC));
}
int main() {
  for (int blocks : {1, 128}) for (int threads : {256, 512}) {
    run("registers only", lab<0>, threads, blocks);
    run("B from LDS", lab<1>, threads, blocks);
    run("B from LDS, pinned", lab<1 | 4>, threads, blocks);
    run("A from global memory", lab<2>, threads, blocks);
    run("A from global memory, pinned", lab<2 | 4>, threads, blocks);
    run("A + B, compiler's order", lab<3>, threads, blocks);
    run("A + B, pinned (the kernel's form)", lab<3 | 4>, threads, blocks);
    run("A + B, pinned, four chains", lab<3 | 4 | 8>, threads, blocks);
    run("A + B, requests spread behind single MFMAs", lab<3 | 16>, threads, blocks);
    run("A + B, spread, four chains", lab<3 | 8 | 16>, threads, blocks);
  }
  return 0;
}
